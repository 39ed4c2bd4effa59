// Hand-written HIP kernels of the EKF-SLAM step for MI355X (gfx950, wave64).
//
// The covariance is symmetric and only its upper triangle is kept, as
//     P(a, b) = P_base[a][b] + sum_k W[a][k] V[k][b] + [a == b < 3] dacc[a]      (a <= b),   P(b, a) := P(a, b)
// with up to KTOT pending ranks: every step appends 2 ranks per observed landmark (the m sequential updates
// of src/replay_no_ros.py:436-480; with P symmetric K_j = (H_j P_j)^T S_j^-1, so V = H_j P_j, W = -K_j come
// from one recurrence) and the O(n^2) pass over P_base ("flush") is paid once per few steps instead of
// (2+m) dense n x n GEMMs per step.  The prediction G_F P G_F^T + F^T R F (:430) changes rows 0,1 of the
// stored triangle only and is applied to P_base in place.
//   k_solve    sequential part on the compressed c x c system (c = 3+2m): gathers the CURRENT P[C,C]
//              (base + pending ranks), one wave runs the recurrences
//   k_panels   thread i replays the step on P(C, i), the column of the current P at its own state index:
//              prediction (rows 0,1 of P_base updated in place), m rank-2 down-dates; appends
//              V = stacked H_j P_j (rank-major, coalesced) and W = -stacked K_j (MFMA-tiled); mean update
//   k_flush    P_base <- P_base + W V + diag(dacc) on the upper triangle: one streaming read-modify-write
//              with the rank-K product on the fp64 matrix cores (v_mfma_f64_16x16x4_f64)
//   k_mirror   lower triangle <- upper (before a download or the dense product)
// With nothing observed and nothing pending, k_predict_rc touches rows 0,1 of P only (O(n)).
// The algebra is restated on the CPU in oracle/ekf_oracle.py::DeferredSymmetricFilter.
#include <algorithm>
#include <type_traits>

#include "ekf_device.h"

// Diagnostic stamps (tools/solve_probe.hip builds with -DEKF_STAMPS); compiled out of the product.
#ifdef EKF_STAMPS
#define STAMP(o, i)                                                                          \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    unsigned long long t_;                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    if (threadIdx.x == 0) (o).stamps[i] = t_;                                                \
  } while (0)
#else
#define STAMP(o, i) do { } while (0)
#endif

#include "ekf_devfn.h"
#include "ekf_host_plan.h"

namespace ekf {

constexpr int PCS = CMAX + 2;   // LDS row stride 37 doubles: column reads by 32 lanes are conflict-free
constexpr int RCH = 8;          // rows per batch of the down-date
constexpr int CPAD = (CMAX + RCH - 1) / RCH * RCH;   // 40
constexpr int WCS = KTOT + 1;   // row stride of the staged W[C,:] (81 doubles: per-lane rows are conflict-free)

// Stage the pending factors restricted to C into LDS: Wc[a][k] = W[C[a]][k], Vc[k][a] = V[k][C[a]]; with
// `fac` they are also written out compactly (facW[a][k], facV[a][k], zero-filled to a multiple of 8 ranks) for
// the panel kernel.  Split in two so that the caller can compute under the gathers' latency: `issue` puts
// all loads of the thread in flight (ranks [kofs, kofs+64)), `commit` stores them.
constexpr int SQ = (CMAX + 2) / 3;                     // rows per wave with three staging waves (the fourth runs the motion model)
struct StageRegs {
  double wv[SQ], vv[SQ];
};
// Thread (wave w < 3, lane) takes rank k = kofs + lane of the rows a = w, w + 3, ...  Every load is one buffer
// instruction: 32-bit lane offset = rank part, scalar offset = row part (C[a] sits in lane a of the caller's Cl);
// nothing sits under a per-lane condition (lanes beyond the pending ranks read the last pending one again -- one
// more lane on a line that is fetched anyway -- and are discarded at commit).
__device__ __forceinline__ void stage_issue(StageRegs& R, __amdgpu_buffer_rsrc_t rsV, __amdgpu_buffer_rsrc_t rsW,
                                            int Cl, int c, int kb, int kofs, int ld, int wave, int lane) {
  const int kc = min(kofs + lane, kb - 1);             // (lanes beyond the pending ranks share the last one's line)
  const unsigned kW = (unsigned)(((kc >> 2) * (ld >> 4)) * 64 + (kc & 3) * 16) * 8u;   // wm_index = rank part + row part
  const unsigned kV = (unsigned)(kc * ld) * 8u;
#pragma unroll
  for (int q = 0; q < SQ; ++q) {
    const int a = wave + 3 * q;
    if (a < c) {                                       // (wave-uniform; a scattered load costs the CU ~1 cycle per line)
      const int row = __builtin_amdgcn_readlane(Cl, a);
      R.wv[q] = ldb8(rsW, kW, (unsigned)((row >> 4) * 64 + (row & 15)) * 8u);
      R.vv[q] = ldb8(rsV, kV, (unsigned)row * 8u);
    } else {
      R.wv[q] = 0.0;
      R.vv[q] = 0.0;
    }
  }
}
// Where solve_body keeps the factors at C in LDS: its own padded arrays (k_solve: conflict-free for the stage writes
// and for the MFMA fragment reads), or the panel kernels' [part][a][k] array (k_panels<.., SPLIT>: the solve workgroup
// goes on to its panel with the factors already in place).
struct FacStd {
  double (*Wc)[WCS];
  double (*Vc)[PCS];
  static constexpr int ROWS = CMAX;
  __device__ __forceinline__ double& w(int a, int k) const { return Wc[a][k]; }
  __device__ __forceinline__ double& v(int k, int a) const { return Vc[k][a]; }
};
template <int CC>
struct FacPanel {
  double (*sF)[CC][KTOT];
  static constexpr int ROWS = CC;
  __device__ __forceinline__ double& w(int a, int k) const { return sF[0][a][k]; }
  __device__ __forceinline__ double& v(int k, int a) const { return sF[1][a][k]; }
};
template <class Fac>
__device__ __forceinline__ void stage_commit(const StageRegs& R, int c, int kb, int kofs, int wave, int lane,
                                             const Fac& F, double* __restrict__ fac) {
  const int k = kofs + lane;
  const int k8 = (kb + 7) & ~7;
  if (k >= KTOT) return;
#pragma unroll
  for (int q = 0; q < SQ; ++q) {
    const int a = wave + 3 * q;                        // (a < 3 SQ = 36 <= PCS: rows c.. and ranks kb.. are staged as zeros,
    if (a < Fac::ROWS) {                               //  so the product below needs no masks)
      const bool in = a < c && k < kb;
      const double w = in ? R.wv[q] : 0.0, v = in ? R.vv[q] : 0.0;
      F.w(a, k) = w;
      F.v(k, a) = v;
      if (fac && a < c && k < k8) {
        fac[a * KTOT + k] = w;
        fac[CMAX * KTOT + a * KTOT + k] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// solve_body: the sequential part of one step for one trajectory, executed by a 256-thread workgroup.
// All four waves gather the current P[C,C] (base + pending ranks); then wave 0 runs the recurrences:
// lane l < c owns compressed index l and keeps column l of P[C,C] in registers (written through to
// LDS for the row/column reads of the other lanes); wave 1 linearises the next landmark beside the
// down-date, which waves 0, 2 and 3 share by rows.  The chain is latency-bound (a lone wave issues
// one fp64 VALU op per ~8 cycles): every phase is written as batches of independent operations.
// Reads mu_in / dacc_in, and -- only where `writer` -- writes mu_out[C], dacc_out, the flags and the
// global SolveOut header (the mean and the pending noise are double-buffered so that no workgroup
// of a step reads an entry another workgroup of the same step writes).  `its` receives the
// per-landmark records.
// ---------------------------------------------------------------------------------------------
struct SolveCore {
  double Pc[CPAD][PCS];
  double Ms[CPAD][PCS];                                // W[C,:] V[:,C], the pending ranks' share of the gathered block
  double2 hpS[CPAD], kcS[CPAD];
  int Cs[CPAD];
  double mot[6];                                       // motion model (wave 3): predicted x, y, theta, G[0,2], G[1,2]
  double2 hS[6];                                       // next linearisation: {h[0][k], h[1][k]} k<5
};
struct SolveLds : SolveCore {
  double Wc[CMAX][WCS];
  double Vc[KTOT][PCS];
  __device__ __forceinline__ FacStd fac_view() { return FacStd{Wc, Vc}; }
};

template <class Fac>
__device__ __forceinline__ void solve_body(SolveCore& L, const Fac& F, const double* __restrict__ Pb,
                                           const double* __restrict__ Vb, const double* __restrict__ Wb,
                                           const double* __restrict__ dacc_in, double* __restrict__ dacc_out,
                                           const double* __restrict__ mu_in_b, double* __restrict__ mu_out_b,
                                           const StepIn& s, SolveOut& o, SolveIter* its, unsigned* flag_b,
                                           double* __restrict__ fac_b, const DeviceConfig& cfg, int ld, int kbase,
                                           bool writer, int neff_eff) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  auto& Pc = L.Pc;
  auto& hpS = L.hpS;
  auto& kcS = L.kcS;
  auto& Cs = L.Cs;
  auto& hS = L.hS;

  STAMP(o, 0);
  // inputs: the index list is fetched unconditionally so that it travels with flags/m (one round trip)
  const int my_idx = (lane >= 3 && lane < CMAX) ? s.idx[(lane - 3) >> 1] : 0;
  const bool do_pred = (s.flags & FLAG_PREDICT) != 0;
  int m = ((s.flags & FLAG_UPDATE) && cfg.enable_measurement_model) ? s.m : 0;
  if (m > MMAX) m = MMAX;
  const int c = 3 + 2 * m;
  const bool on = lane < c;
  const int ll = on ? lane : 0;                       // clamped lane for in-bounds LDS reads
  const int Cl = (lane < 3) ? lane : (on ? 3 + 2 * my_idx + ((lane - 3) & 1) : 0);
  if (tid < CPAD) {
    Cs[tid] = Cl;
    kcS[tid] = make_double2(0.0, 0.0);
    hpS[tid] = make_double2(0.0, 0.0);
    if (writer && tid < CMAX + 1) o.C[tid] = Cl;
  }
  const double mu_l = mu_in_b[Cl];
  // pose-block noise pending since the last covariance pass; with no rank pending there is none (the buffers are
  // not cleared after a pass: that would be two more launches per pass)
  const double d0 = kbase > 0 ? dacc_in[0] : 0.0, d1 = kbase > 0 ? dacc_in[1] : 0.0, d2 = kbase > 0 ? dacc_in[2] : 0.0;
  const double lin_in = s.lin, ang_in = s.ang;         // (fetched with the other inputs: one round trip)
  int cmax = Cl;                                       // largest gathered index (wave-wide maximum)
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) cmax = max(cmax, __shfl_xor(cmax, sh));
  WG_LDS_BARRIER();
  STAMP(o, 1);
  // current P[C,C] = P_base[C,C] + W[C,:] V[:,C] + diag(dacc): the base loads are issued first ...
  // (thread = (wave w, lane): column C[lane] of rows w, w+4, ...: no integer division on the path)
  constexpr int GQ = (CMAX + 3) / 4;                   // 9 rows per wave at most
  const int gw = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (lane a of every wave holds C[a]: the row indices come through the scalar registers; idle lanes read entry
  //  (0, C[r]) and discard it, so that no load sits under a per-lane condition)
  double gv[GQ];
#pragma unroll
  for (int q = 0; q < GQ; ++q) {
    const int r = gw + 4 * q;
    gv[q] = 0.0;
    if (r < c) {                                       // (wave-uniform)
      const int Cr = __builtin_amdgcn_readlane(Cl, r);
      gv[q] = Pb[p_index(ld, min(Cr, Cl), max(Cr, Cl))];                       // the upper triangle is authoritative
    }
  }
  STAMP(o, 110);
  // ... then waves 0-2 gather the pending factors at C while wave 3 runs the motion model
  // (src/replay_no_ros.py:368-417; two sincos, a division and a wrap: ~1900 cycles of a lone wave)
  StageRegs SR;
  const __amdgpu_buffer_rsrc_t rsV = rs_rsrc(Vb), rsW = rs_rsrc(Wb);
  if (gw < 3) {
    if (kbase > 0) stage_issue(SR, rsV, rsW, Cl, c, kbase, 0, ld, gw, lane);
  } else {
    const double th = __shfl(mu_l, 2);
    double g0 = 0.0, g1 = 0.0, nx = __shfl(mu_l, 0), ny = __shfl(mu_l, 1), nth = th;
    if (do_pred && !cfg.disable_motion_model) {
      const double lin = lin_in, ang = ang_in;
      double s0, c0;
      sincos(th, &s0, &c0);
      if (cfg.enable_circular_interpolation && fabs(ang) > cfg.arc_threshold) {   // :390 arc
        double s1, c1;
        sincos(th + ang, &s1, &c1);
        const double r = lin / ang;
        nx += -r * s0 + r * s1;
        ny += r * c0 - r * c1;
        nth = wrap_pi(th + ang);                            // :397
        g0 = -r * c0 + r * c1;                              // :401
        g1 = -r * s0 + r * s1;                              // :402
      } else {                                              // :376 straight / :405-417 linear mode
        nx += lin * c0;
        ny += lin * s0;
        if (!cfg.enable_circular_interpolation) nth = th + ang;   // no wrap (:409); :381 keeps theta
        g0 = -lin * s0;
        g1 = lin * c0;
      }
    }
    if (lane == 0) {
      L.mot[0] = nx;
      L.mot[1] = ny;
      L.mot[2] = nth;
      L.mot[3] = g0;
      L.mot[4] = g1;
    }
  }
  STAMP(o, 112);
  if (kbase > 0) {
    if (gw < 3) {
      stage_commit(SR, c, kbase, 0, gw, lane, F, writer ? fac_b : nullptr);
      if (kbase > 64) {                                // (more than 64 pending ranks: second pass)
        stage_issue(SR, rsV, rsW, Cl, c, kbase, 64, ld, gw, lane);
        stage_commit(SR, c, kbase, 64, gw, lane, F, writer ? fac_b : nullptr);
      }
    }
    WG_LDS_BARRIER();
  }
  STAMP(o, 113);
  // pending ranks: M = W[C,:] V[:,C] (c x c, 16x16 tiles dealt to the four waves, v_mfma_f64_16x16x4 with both
  // operands straight from the staged factors, four k-tiles of fragments fetched per round trip to LDS), parked in
  // Pc; entry (r, l) of the gathered block then takes M[r][l] where P(C[r], C[l]) is stored that way round and
  // M[l][r] where it is stored mirrored
  double mcor[GQ];
#pragma unroll
  for (int q = 0; q < GQ; ++q) mcor[q] = 0.0;
  if (kbase > 0) {
    const int T = (c + 15) >> 4, nkt = (kbase + 3) >> 2;
    const int li = lane & 15, lq = lane >> 4;
    for (int t = gw; t < T * T; t += 4) {
      const int rt = (t >= 2 * T) ? 2 : ((t >= T) ? 1 : 0), ct = t - rt * T;   // T <= 3
      const int ar = 16 * rt + li, bc = 16 * ct + li;
      // Operands straight from the staged factors: A[i = li][k = lq] = Wc[ar][4 kt + lq], B[k = lq][j = li] =
      // Vc[4 kt + lq][bc].  Rows / columns beyond c give rows / columns of the product nobody stores; ranks beyond
      // the pending ones were staged as zeros.  Four k-tiles of fragments per round trip to LDS, the next four in
      // flight under the MFMAs; even and odd k-tiles accumulate separately (a dependent v_mfma_f64_16x16x4 waits
      // for its predecessor).
      const int ra = min(ar, Fac::ROWS - 1), cb = min(bc, Fac::ROWS - 1);
      const int nb = (nkt + 3) >> 2;                   // batches of 4 k-tiles (16 ranks); 16 nb <= KTOT
      double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
      double av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = F.w(ra, lq + 4 * u);
        bv[u] = F.v(lq + 4 * u, cb);
      }
      for (int bt = 0; bt < nb; ++bt) {
        const int kn = min(16 * (bt + 1), KTOT - 16);  // (the fetch behind the last batch is not used)
        double an[4], bn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          an[u] = F.w(ra, lq + kn + 4 * u);
          bn[u] = F.v(lq + kn + 4 * u, cb);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          av[u] = an[u];
          bv[u] = bn[u];
        }
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = 16 * rt + lq + 4 * reg, l = 16 * ct + li;
        if (r < c && l < c) L.Ms[r][l] = acc0[reg] + acc1[reg];
      }
    }
    WG_LDS_BARRIER();
    STAMP(o, 114);
#pragma unroll
    for (int q = 0; q < GQ; ++q) {
      const int r = gw + 4 * q;
      if (r < c) {                                     // (wave-uniform)
        const int Cr = __builtin_amdgcn_readlane(Cl, r);
        const double m_rl = L.Ms[r][ll], m_lr = L.Ms[ll][r];
        mcor[q] = (Cr <= Cl) ? m_rl : m_lr;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < GQ; ++q) {
    const int r = gw + 4 * q;
    if (on && r < c) {
      double v = gv[q] + mcor[q];
      if (q == 0 && r == lane && r < 3) v += (r == 0) ? d0 : ((r == 1) ? d1 : d2);   // (r >= 4 for q >= 1)
      Pc[r][lane] = v;
    }
  }
  WG_LDS_BARRIER();
  // (every wave) what wave 3 made of the motion model
  const double g0 = L.mot[3], g1 = L.mot[4];
  const double mu_pred = (lane < 3) ? L.mot[lane] : mu_l;
  // The down-date of one landmark, P[r][l] -= K[r,:] . (H P)[:, l] (:480), on the rows slot, slot + 3, ... of the
  // block in LDS, column l = lane: waves 0, 2 and 3 each take a third of the rows (a lone wave issues one fp64
  // operation per ~8 cycles: 2 c of them in a row were the longest phase of an iteration).  Rows >= c have K = 0.
  auto downdate_rows = [&](int slot, double2 hp) {
#pragma unroll
    for (int q0 = 0; q0 < 14; q0 += 7) {
      if (slot + 3 * q0 < c) {                         // (uniform) 21 rows per batch over the three waves
        double2 kr[7];
        double pv[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) {
          const int r = min(slot + 3 * (q0 + u), CPAD - 1);
          kr[u] = kcS[r];
          pv[u] = Pc[r][ll];
        }
#pragma unroll
        for (int u = 0; u < 7; ++u) pv[u] = fma(-kr[u].x, hp.x, pv[u]);
#pragma unroll
        for (int u = 0; u < 7; ++u) pv[u] = fma(-kr[u].y, hp.y, pv[u]);
        if (on) {
#pragma unroll
          for (int u = 0; u < 7; ++u) {
            const int r = slot + 3 * (q0 + u);
            if (r < CPAD) Pc[r][lane] = pv[u];
          }
        }
      }
    }
  };
  if (tid >= 64) {
    // Helper waves.  Wave 1 owns the mean: it publishes the predicted mean and linearises landmark 0 while wave 0
    // forms the predicted covariance block; in iteration j it adds K_j y_j (:476) as soon as wave 0 has published
    // K_j, forms the Jacobian of landmark j+1 at the new mean (~350 cycles of a lone wave) while waves 0, 2 and 3
    // down-date, and computes the innovation of landmark j+1 (atan2, wrap: ~600 cycles) behind the barrier, under
    // wave 0's next (H P), S and K -- only the mean needs it.
    const int hw = tid >> 6;
    double mu_cur = mu_pred, y0 = 0.0, y1 = 0.0;
    // (wave 1) lane j holds the measurement of landmark j: fetched once, up front
    const double zr = s.range[lane & (MMAX - 1)], zb = s.bearing[lane & (MMAX - 1)];
    auto jacobian_at_mean = [&](int a) -> LinGeom {    // mean entries of the pose and of landmark (a - 3) / 2 from the lanes that own them
      double hn[2][5];
      const LinGeom g = linearize_h(read_lane(mu_cur, 0), read_lane(mu_cur, 1), read_lane(mu_cur, 2), read_lane(mu_cur, a),
                                    read_lane(mu_cur, a + 1), hn);
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) hS[k] = make_double2(hn[0][k], hn[1][k]);
      }
      return g;
    };
    if (hw == 1) {
      LinGeom g{};
      if (m > 0) g = jacobian_at_mean(3);
      WG_LDS_BARRIER();                                // S0: predicted covariance block and hS published
      if (m > 0) innovation(g, read_lane(zr, 0), read_lane(zb, 0), y0, y1);
    } else {
      WG_LDS_BARRIER();                                // S0
    }
    for (int j = 0; j < m; ++j) {
      WG_LDS_BARRIER();                                // b1(j): K_j and (H P) of landmark j are in LDS
      if (hw == 1) {
        const double2 kj = kcS[ll];
        if (on) mu_cur += kj.x * y0 + kj.y * y1;      // :476
        if (lane == 0) *reinterpret_cast<double2*>(its[j].y) = make_double2(y0, y1);
        LinGeom g{};
        if (j + 1 < m) g = jacobian_at_mean(3 + 2 * (j + 1));
        WG_LDS_BARRIER();                              // b2(j): hS ready, covariance block down-dated
        if (j + 1 < m) innovation(g, read_lane(zr, j + 1), read_lane(zb, j + 1), y0, y1);
      } else {
        if (j + 1 < m) downdate_rows(hw - 1, hpS[ll]);   // waves 2, 3: their third of the rows
        WG_LDS_BARRIER();                              // b2(j)
      }
    }
    if (hw == 1) {
      bool bad = false;
      if (on && writer) {
        mu_out_b[Cl] = mu_cur;
        bad = !(fabs(mu_cur) <= 1.79769313486231570815e308);
      }
      if (__any(bad) && lane == 0) atomicOr(flag_b, EKF_FLAG_NONFINITE);
    }
  } else {
  STAMP(o, 2);
  const double rd0 = do_pred ? cfg.rd[0] : 0.0, rd1 = do_pred ? cfg.rd[1] : 0.0,
               rd2 = do_pred ? cfg.rd[2] : 0.0;
  STAMP(o, 3);
  // P'[C,C] = Gc P[C,C] Gc^T + Rt  (:428-430 restricted to C).  Only rows 0,1 and columns 0,1 of the block change
  // (row ops X = Gc P on rows 0,1 with row 2, then column ops X Gc^T on columns 0,1 with column 2 of X), and the
  // gathered block is exactly symmetric: lane r holds P[0..2][r] = P[r][0..2] and produces P'[r][0], P'[r][1],
  // which for r >= 2 are also P'[0][r], P'[1][r].
  const double p0 = on ? Pc[0][ll] : 0.0, p1 = on ? Pc[1][ll] : 0.0, p2 = on ? Pc[2][ll] : 0.0;
  const double s20 = __shfl(p2, 0), s21 = __shfl(p2, 1), p22 = __shfl(p2, 2);
  if (writer && lane < CMAX + 1) {                     // rows 0,1 of the gathered block before the step: the panel
    o.prow[0][lane] = p0;                              // kernel's state indices 0,1 start from these (their own
    o.prow[1][lane] = p1;                              // gather would race with the row update of other columns)
  }
  {
    const double gr = (lane == 0) ? g0 : ((lane == 1) ? g1 : 0.0);
    double x0 = p0, x1 = p1, x2 = p2;                  // row r of X, columns 0..2
    if (lane < 2) {
      x0 = fma(gr, s20, p0);
      x1 = fma(gr, s21, p1);
      x2 = fma(gr, p22, p2);
    }
    double c0n = fma(g0, x2, x0), c1n = fma(g1, x2, x1);
    if (lane == 0) c0n += rd0;
    if (lane == 1) c1n += rd1;
    if (on) {
      Pc[lane][0] = c0n;
      Pc[lane][1] = c1n;
      if (lane >= 2) {
        Pc[0][lane] = c0n;
        Pc[1][lane] = c1n;
      }
      if (lane == 2) Pc[2][2] = p2 + rd2;
    }
  }
  if (lane == 0 && writer) {
    o.cmax = cmax;
    o.g[0] = g0;
    o.g[1] = g1;
    o.rd[0] = rd0;
    o.rd[1] = rd1;
    o.rd[2] = rd2;
    o.p22h = 0.5 * p22;
    o.dacc_old[0] = d0;
    o.dacc_old[1] = d1;
    o.dacc_old[2] = d2;
    o.c = c;
    o.m = m;
    o.kbase = kbase;
    o.neff = neff_eff;
    dacc_out[0] = d0 + rd0;                            // the pose-block noise joins the pending update
    dacc_out[1] = d1 + rd1;
    dacc_out[2] = d2 + rd2;
  }
  WG_LDS_BARRIER();                                    // S0 (helper waves wait here too)

  STAMP(o, 4);
  // ---- sequential per-landmark recurrences (:436-480) on the compressed system ----
  double h[2][5];
  if (m > 0) {                                         // landmark 0 was linearised by wave 1
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const double2 t = hS[k];
      h[0][k] = t.x;
      h[1][k] = t.y;
    }
  }
  STAMP(o, 5);
  for (int j = 0; j < m; ++j) {
    const int a = 3 + 2 * j;
    SolveIter& it = its[j];
    STAMP(o, 8 + 6 * j);
    // phase A: rows sel of P_j at column C[lane] give (H P)[:, C[lane]]; P is symmetric (only its upper
    // triangle is stored), so P H^T is the transpose and the gain needs no second product.
    // The instruction count is what matters for a lone wave: S is formed from the five hp pairs the
    // other lanes publish in LDS (20 FMAs), not recomputed from the 5x5 block.
    double pr[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) pr[k] = Pc[(k < 3) ? k : a + (k - 3)][ll];
    double hp0 = h[0][0] * pr[0], hp1 = h[1][0] * pr[0];
#pragma unroll
    for (int k = 1; k < 5; ++k) {
      hp0 = fma(h[0][k], pr[k], hp0);
      hp1 = fma(h[1][k], pr[k], hp1);
    }
    if (on) hpS[lane] = make_double2(hp0, hp1);
    WAVE_LDS_SYNC();
    STAMP(o, 9 + 6 * j);
    // phase B: S = H P H^T + Q (:473), every lane redundantly
    double2 hv[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) hv[k] = hpS[(k < 3) ? k : a + (k - 3)];
    double S00 = cfg.qd[0], S01 = 0.0, S10 = 0.0, S11 = cfg.qd[1];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      S00 = fma(hv[k].x, h[0][k], S00);
      S01 = fma(hv[k].x, h[1][k], S01);
      S10 = fma(hv[k].y, h[0][k], S10);
      S11 = fma(hv[k].y, h[1][k], S11);
    }
    const double rdet = 1.0 / (S00 * S11 - S01 * S10);
    const double i00 = S11 * rdet, i01 = -S01 * rdet, i10 = -S10 * rdet, i11 = S00 * rdet;
    const double k0 = hp0 * i00 + hp1 * i10;                            // K_j[C[lane], :] = (H P)[:, C[lane]]^T S^-1
    const double k1 = hp0 * i01 + hp1 * i11;
    if (on) kcS[lane] = make_double2(k0, k1);                           // (wave 1 adds K_j y_j to the mean, :476)
    if (lane < CMAX) *reinterpret_cast<double2*>(it.kc[lane]) = on ? make_double2(k0, k1) : make_double2(0.0, 0.0);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 5; ++k) *reinterpret_cast<double2*>(it.h5t[k]) = make_double2(h[0][k], h[1][k]);
      *reinterpret_cast<double2*>(&it.si[0]) = make_double2(i00, i01);
      *reinterpret_cast<double2*>(&it.si[2]) = make_double2(i10, i11);
    }
    WG_LDS_BARRIER();                                  // b1(j): wave 1 starts the next linearisation
    STAMP(o, 10 + 6 * j);
    // phase C: down-date (:480), this wave's third of the rows (waves 2, 3 take the others, wave 1 linearises
    // landmark j+1 meanwhile)
    STAMP(o, 11 + 6 * j);
    if (j + 1 < m) downdate_rows(0, make_double2(hp0, hp1));
    WG_LDS_BARRIER();                                  // b2(j)
    if (j + 1 < m) {
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const double2 t = hS[k];
        h[0][k] = t.x;
        h[1][k] = t.y;
      }
    }
    STAMP(o, 12 + 6 * j);
  }
  STAMP(o, 6);
  }   // wave 0
}

__global__ __launch_bounds__(256) void k_solve(const double* __restrict__ P, const double* __restrict__ V,
                                               const double* __restrict__ W,
                                               const double* __restrict__ dacc_in,
                                               double* __restrict__ dacc_out,
                                               const double* __restrict__ mu_in,
                                               double* __restrict__ mu_out,
                                               const int* __restrict__ nact,
                                               const StepIn* __restrict__ in,
                                               SolveOut* __restrict__ out,
                                               unsigned* __restrict__ flags, double* __restrict__ fac,
                                               const int* __restrict__ neff_floor,
                                               unsigned* __restrict__ queue,
                                               DeviceConfig cfg, int ld, long pstride, int kbase) {
  __shared__ SolveLds L;
  const int b = blockIdx.x;
  // the work-queue heads of the row-slab covariance pass start every pass at zero: some step precedes every pass
  if (b == 0 && threadIdx.x < 8) queue[threadIdx.x * RS_QSTRIDE] = 0u;
  // active bound of this step: what the host baked into the record, raised to the handle's floor (the bound the
  // state had when the enqueueing call started: a stream uploaded earlier knows only its own observations)
  const int neff_eff = min(nact[b], max(in[b].neff, neff_floor[b]));
  solve_body(L, L.fac_view(), P + (long)b * pstride, V + (long)b * KTOT * ld, W + (long)b * KTOT * ld, dacc_in + 4 * b,
             dacc_out + 4 * b, mu_in + (long)b * ld, mu_out + (long)b * ld, in[b], out[b], out[b].it,
             flags + b, fac + (long)b * FACS, cfg, ld, kbase, true, neff_eff);
}

// ---------------------------------------------------------------------------------------------
// k_panels: thread i replays the step on x[a] = P(C[a], i), the column of the current covariance at its
// own state index restricted to the gathered rows C (P is symmetric, only the upper triangle of P_base
// is stored):
//   gather     x[a] = P_base(min(C[a],i), max(C[a],i)) + pending ranks:  sum_k W[C[a]][k] V[k][i]  where the
//              entry is stored as (C[a], i), else  sum_k W[i][k] V[k][C[a]]   (its mirror);  V[:,i] and W[i,:]
//              are coalesced vector loads (8 ranks at a time, the next 8 in flight under the FMAs)
//   predict    P' = G_F P G_F^T + F^T R F on the panel (src/replay_no_ros.py:430); of the stored triangle it
//              changes rows 0,1 only, and thread i adds its two entries to P_base in place
//   m updates  u = (H_j P_j)[:, i] = h5 . x[sel];  K_j[i,:] = u^T S_j^-1;  x -= K_j[C,:] u   (:473-480)
//              appended ranks  V[kb+2j..][i] = u,  W[i][kb+2j..] = -K_j[i,:];  mean += K_j[i,:] y_j  (:476)
// A workgroup is NW independent waves of 64 state indices.  What they share (the per-landmark records of
// k_solve and its compact copy of the factors at C) is staged into LDS once, with one round of loads, and
// then read as 16-byte broadcasts; after the single barrier the waves never synchronise again.
// ---------------------------------------------------------------------------------------------
template <int MCAP, int NW, bool KSPLIT>
struct PanelLds {
  static constexpr int CC = 3 + 2 * MCAP;
  __attribute__((aligned(16))) double sF[2][CC][KTOT];   // [0]: W[C[a]][k], [1]: V[k][C[a]]
  SolveIter sIt[MCAP];
  int sC[CC + 1];
};
template <int MCAP, int NPART>
struct PanelPartLds {
  double sPart[NPART][3 + 2 * MCAP][64];               // rank-split shapes: partial gathers of the other waves
};

// ---- the gather of the pending ranks, 8 ranks at a time ----
// The loads of a group: V[k][i] (coalesced along the state indices) and, where some gathered index lies beyond the
// wave's first state index (NEEDW), W[i][k].  No load sits under a branch -- the group after the last one is "fetched"
// too, but pointed at the last valid group again (rows that were just read, not rank slots nobody needs: a V row costs
// a trip to HBM) and masked -- so the compiler can count the outstanding loads and the next group really is in flight
// under this group's FMAs.
template <bool NEEDW>
__device__ __forceinline__ void pg_load(const double* __restrict__ vlane, const double* __restrict__ wlane, int ld,
                                        int ld16, int kb, int k0, double (&vv)[8], double (&ww)[8]) {
  const int k_last = ((kb - 1) >> 3) << 3;
  const int kc = min(k0, k_last);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const double x = vlane[(long)(kc + u) * ld];
    vv[u] = (k0 + u < kb) ? x : 0.0;
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (NEEDW) {
      const double x = wlane[((long)((kc + u) >> 2) * ld16) * 64 + ((kc + u) & 3) * 16];
      ww[u] = (k0 + u < kb) ? x : 0.0;
    } else {
      ww[u] = 0.0;
    }
  }
}
// X[a] += sum over the group's ranks of W[C[a]][k] V[k][i] where the entry is stored as (C[a], i), of W[i][k] V[k][C[a]]
// where it is stored mirrored.  Four gathered rows at a time: their 16 coefficient reads (16-byte LDS broadcasts) are in
// flight together, and each row's eight FMAs run as two independent chains (the branch per row is wave-uniform).
template <int CC>
__device__ __forceinline__ void pg_accumulate(double (&X)[CC], const double (&v)[8], const double (&w)[8],
                                              const double (*sF)[CC][KTOT], const int* sC, int k0, int i0, int ii) {
#pragma unroll
  for (int g = 0; g < CC; g += 4) {
    double cf[4][8];                                   // (plain doubles: arrays of double2 end up in scratch)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int a = min(g + q, CC - 1);                // (rows c.. of a short step hold stale, unused data)
      const double2* src = reinterpret_cast<const double2*>(__builtin_assume_aligned(&sF[sC[a] <= i0 ? 0 : 1][a][k0], 16));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double2 t = src[u];
        cf[q][2 * u] = t.x;
        cf[q][2 * u + 1] = t.y;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int a = g + q;
      if (a < CC) {
        const int row = sC[a];
        double s0 = X[a], s1 = 0.0;
        if (row <= i0) {                               // stored as (C[a], i) for the whole wave
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            s0 = fma(cf[q][2 * u], v[2 * u], s0);
            s1 = fma(cf[q][2 * u + 1], v[2 * u + 1], s1);
          }
        } else if (row > i0 + 63) {                    // mirrored for the whole wave
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            s0 = fma(w[2 * u], cf[q][2 * u], s0);
            s1 = fma(w[2 * u + 1], cf[q][2 * u + 1], s1);
          }
        } else {                                       // the wave straddles C[a]: both forms, chosen per lane
          const double2* wa = reinterpret_cast<const double2*>(__builtin_assume_aligned(&sF[0][a][k0], 16));
          const bool up = row <= ii;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const double2 fw2 = wa[u];
            s0 = fma(up ? fw2.x : w[2 * u], up ? v[2 * u] : cf[q][2 * u], s0);
            s1 = fma(up ? fw2.y : w[2 * u + 1], up ? v[2 * u + 1] : cf[q][2 * u + 1], s1);
          }
        }
        X[a] = s0 + s1;
      }
    }
  }
}

// What every shape of the panel code needs to know about its 64 state indices.
struct PanelIdx {
  int lane, i0, i, ii, ld, ld16, n;
  bool act;
  double *Pb, *Vb, *Wb;
  const double* mu_in_b;
  double* mu_out_b;
};
__device__ __forceinline__ PanelIdx panel_idx(double* P, double* V, double* W, const double* mu_in, double* mu_out,
                                              int ld, long pstride, int b, int n, int i0) {
  PanelIdx t;
  t.lane = threadIdx.x & 63;
  t.i0 = i0;
  t.i = i0 + t.lane;
  t.act = t.i < n;
  t.ii = t.act ? t.i : n - 1;                          // idle lanes shadow the last state index (no stores)
  t.ld = ld;
  t.ld16 = ld >> 4;
  t.n = n;
  t.Pb = P + (long)b * pstride;
  t.Vb = V + (long)b * KTOT * ld;
  t.Wb = W + (long)b * KTOT * ld;
  t.mu_in_b = mu_in + (long)b * ld;
  t.mu_out_b = mu_out + (long)b * ld;
  return t;
}
// x[a] = P_base(min(C[a], i), max(C[a], i)): the base entries of the gather
template <int CC>
__device__ __forceinline__ void panel_base_gather(const PanelIdx& t, const int* Crow, double (&X)[CC]) {
#pragma unroll
  for (int a = 0; a < CC; ++a) {
    const int row = Crow[a];
#ifdef PANELS_SKIP_COLG                                 /* diagnostic build: no column-direction gathers */
    X[a] = t.Pb[p_index(t.ld, min(row, t.i0), max(row, t.ii))];
#else
    X[a] = t.Pb[p_index(t.ld, min(row, t.ii), max(row, t.ii))];
#endif
  }
}
// Beyond the active bound the rows and columns of P are exactly zero off the diagonal: this step's ranks are zero
// there and the mean is carried over.
template <int MCAP>
__device__ __forceinline__ void panel_beyond_bound(const PanelIdx& t, int kb) {
  constexpr int KTP = ranks_for(MCAP);
  if (!t.act) return;
  for (int k = kb; k < ((kb + KTP + 3) & ~3); ++k) {
    t.Vb[(long)k * t.ld + t.i] = 0.0;
    t.Wb[wm_index(t.ld16, k, t.i)] = 0.0;
  }
  t.mu_out_b[t.i] = t.mu_in_b[t.i];
}

// From the gathered x[a] = P(C[a], i) on: prediction on the panel, the m sequential rank-2 updates, the new rank
// entries, the mean.  `o`, `sIt` (per-landmark records) and `sC` may live in LDS or (o) in global memory.
template <int MCAP>
__device__ __forceinline__ void panels_finish(const PanelIdx& t, const SolveHead& o, const SolveIter* sIt, const int* sC,
                                              double (&X)[3 + 2 * MCAP]) {
  constexpr int CC = 3 + 2 * MCAP, KTP = ranks_for(MCAP);
  const int lane = t.lane, i0 = t.i0, i = t.i, ii = t.ii, ld = t.ld, ld16 = t.ld16;
  const bool act = t.act;
  double *Vb = t.Vb, *Wb = t.Wb;
  const int kb = o.kbase, m = min(o.m, MCAP), c = o.c;
#pragma unroll
  for (int a = 0; a < 3; ++a)
    if (ii == a) X[a] += o.dacc_old[a];                // pending pose-block noise on the diagonal

  // ---- predict:  P' = G_F P G_F^T + F^T R F  (src/replay_no_ros.py:430) on the panel ----
  // Of the stored triangle it changes rows 0 and 1 only: thread i adds its two entries to P_base directly
  // (the pending ranks are unaffected), so the prediction costs no rank of the covariance pass.
  const double g0 = o.g[0], g1 = o.g[1];
  const double gj = (ii == 0) ? g0 : ((ii == 1) ? g1 : 0.0);
  double d0, d1;                                       // P'(0,i) - P(0,i),  P'(1,i) - P(1,i)
  if (i0 == 0) {
    if (lane < 2) {                                    // rows 0,1 of P_base are being rewritten by the other columns:
#pragma unroll
      for (int a = 0; a < CC; ++a) X[a] = o.prow[lane][a];   // state indices 0,1 take the solve's gather
    }
    const double p22 = __shfl(X[2], 2);
    double col2[CC];                                   // X[:,2] after the row ops, for the column ops of lanes 0,1
#pragma unroll
    for (int a = 0; a < CC; ++a) col2[a] = __shfl(X[a], 2);
    col2[0] += g0 * p22;
    col2[1] += g1 * p22;
    d0 = g0 * X[2];                                    // row ops on rows 0,1
    d1 = g1 * X[2];
    if (ii < 2) {
      d0 += gj * col2[0];
      d1 += gj * col2[1];
#pragma unroll
      for (int a = 2; a < CC; ++a) X[a] += gj * col2[a];
    }
  } else {
    d0 = g0 * X[2];
    d1 = g1 * X[2];
  }
  X[0] += d0;
  X[1] += d1;
  if (act) {
    double* p0 = t.Pb + p_col(ld, i);                  // entry (0, i)
    *p0 += d0;
    if (i >= 1) p0[p_lds(ld)] += d1;                   // entry (1, i); (1, 0) lies below the diagonal
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
    if (a == ii) X[a] += o.rd[a];

  // ---- sequential landmark updates ----
  double dm = 0.0;
#pragma unroll
  for (int it = 0; it < MCAP; ++it) {
    const int kr = kb + 2 * it;
    if (it < m) {
      const SolveIter& I = *reinterpret_cast<const SolveIter*>(__builtin_assume_aligned(sIt + it, 16));
      const int a0 = 3 + 2 * it;
      double2 hk[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) hk[k] = *reinterpret_cast<const double2*>(I.h5t[k]);
      const double2 s01 = *reinterpret_cast<const double2*>(&I.si[0]);
      const double2 s23 = *reinterpret_cast<const double2*>(&I.si[2]);
      const double2 yy = *reinterpret_cast<const double2*>(I.y);
      double e0 = hk[0].x * X[0], e1 = hk[0].y * X[0];   // (H_j P_j)[:, i] = h5 . x[sel]
#pragma unroll
      for (int k = 1; k < 5; ++k) {
        const double xv = (k < 3) ? X[k] : X[a0 + (k - 3)];
        e0 = fma(hk[k].x, xv, e0);
        e1 = fma(hk[k].y, xv, e1);
      }
      const double f0 = e0 * s01.x + e1 * s23.x;       // K_j[i,:] = (H_j P_j)[:, i]^T S^-1  (P symmetric)
      const double f1 = e0 * s01.y + e1 * s23.y;
      dm += f0 * yy.x + f1 * yy.y;
      if (act) {
        Vb[(long)kr * ld + i] = e0;
        Vb[(long)(kr + 1) * ld + i] = e1;
        Wb[wm_index(ld16, kr, i)] = -f0;
        Wb[wm_index(ld16, kr + 1, i)] = -f1;
      }
      if (it + 1 < m) {                                // x[a] -= K_j[C[a],:] . (H_j P_j)[:, i]
        // only the rows a later landmark still reads: the pose rows and the rows of landmarks it+1..
#pragma unroll
        for (int a = 0; a < CC; ++a) {
          if (a < 3 || a >= a0 + 2) {
            const double2 kc = *reinterpret_cast<const double2*>(I.kc[a]);
            X[a] = fma(-kc.x, e0, X[a]);
            X[a] = fma(-kc.y, e1, X[a]);
          }
        }
      }
    } else if (act) {
      Vb[(long)kr * ld + i] = 0.0;
      Vb[(long)(kr + 1) * ld + i] = 0.0;
      Wb[wm_index(ld16, kr, i)] = 0.0;
      Wb[wm_index(ld16, kr + 1, i)] = 0.0;
    }
  }
  if (act) {
    for (int k = kb + KTP; k < ((kb + KTP + 3) & ~3); ++k) {   // k-tile pad
      Vb[(long)k * ld + i] = 0.0;
      Wb[wm_index(ld16, k, i)] = 0.0;
    }
    bool inC = false;
    for (int a = 0; a < c; ++a) inC |= (sC[a] == i);
    if (!inC) t.mu_out_b[i] = t.mu_in_b[i] + dm;       // the solve wrote the entries in C
  }
}

// ---- hand-over of a solve's results to the panel workgroups of the same launch ----
// Workgroups of one launch run on different XCDs, whose L2s are not coherent with each other.  Publishing through
// release/acquire fences works (buffer_wbl2 / buffer_inv), but each fence writes back or invalidates a whole L2: fine
// with 64 workgroups, ruinous with 512 that are streaming their panels through those L2s (k_panels<.., SPLIT>: 80 us per
// step instead of 35).  So the results travel through a mailbox instead: the solve workgroup copies header and records
// (6.4 KB) with device-scope stores (sc1: written through), waits for them, then sets the trajectory's step counter the
// same way; a panel wave polls the counter and reads the mailbox with device-scope loads (sc1: past its own L2).  No
// fence, nothing else leaves or enters a cache.
__device__ __forceinline__ void mailbox_publish(const SolveOut& src, SolveOut* mbox, int m, unsigned* ready, unsigned seq,
                                                int publish) {
  __syncthreads();                                     // every wave's own stores to src have been issued and waited for
  const int words = (int)(sizeof(SolveHead) / 8) + m * (int)(sizeof(SolveIter) / 8);
  const double* s = reinterpret_cast<const double*>(&src);
  double* d = reinterpret_cast<double*>(mbox);
  for (int w = threadIdx.x; w < words; w += blockDim.x)
    __hip_atomic_store(d + w, s[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // written through ...
  __syncthreads();                                     // ... by every wave ...
  if (threadIdx.x == 0 && publish) __hip_atomic_store(ready, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... then the word
}
// One wave: wait (bounded) for the counter, then fetch header and records into LDS.  Returns false on a timeout.
__device__ __forceinline__ bool mailbox_fetch(const SolveOut* mbox, int m, const unsigned* ready, unsigned seq, int spin_limit,
                                              SolveHead* head_lds, SolveIter* it_lds) {
  const int lane = threadIdx.x & 63;
  unsigned got = 0;
  for (int spin = 0; spin < spin_limit; ++spin) {
    got = __hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (got == seq) break;
    __builtin_amdgcn_s_sleep(4);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the loads below are issued after the counter was seen)
  const double* s = reinterpret_cast<const double*>(mbox);
  constexpr int HW = (int)(sizeof(SolveHead) / 8), IW = (int)(sizeof(SolveIter) / 8);
  double* dh = reinterpret_cast<double*>(head_lds);
  double* di = reinterpret_cast<double*>(it_lds);
  for (int w = lane; w < HW; w += 64) dh[w] = __hip_atomic_load(s + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int w = lane; w < m * IW; w += 64) di[w] = __hip_atomic_load(s + HW + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  WAVE_LDS_SYNC();
  return got == seq;
}

constexpr int SPLIT_SPIN_LIMIT = 1 << 18;              // bounded device-scope waits: x (s_sleep + one load from L2), tens of ms

// (k_panels keeps its own copy of the gather and replay code below: routed through the shared helpers above, the
//  same source compiles to a slower kernel at 32 trajectories -- 35.9 us against 33.7 us, same box)
//
// SPLIT (throughput shape only): the whole step in this one launch, as in k_step_split, for batches whose panel
// workgroups leave room on the chip for one more workgroup per trajectory.  Workgroup 0 of a trajectory is the solve
// (its factors at C live in sF) and publishes its results through the mailbox; workgroup x >= 1 takes block x - 1 of
// 256 state indices, forms C from the step's inputs, stages the factors at C for itself, gathers its panel while the
// solve runs, and waits for the trajectory's step counter before it fetches header and records and replays.  Every
// write of a panel workgroup comes after its wait.  (Letting the solve workgroup take a block of panel work after its
// solve -- so that a batch that fills the chip exactly, 32 trajectories at N=2000, could use this too -- was measured:
// 53 us per step against 47 us with two launches; the solve runs slower beside the gathers and its own block then
// starts late.)
struct SplitArgs {
  const double* dacc_in;
  double* dacc_out;
  const StepIn* in;
  SolveOut* out;
  unsigned* flags;
  double* fac;
  const int* neff_floor;
  unsigned* queue;
  SolveOut* mbox;       // per trajectory: header + records of this launch's solve, written through (see mailbox_publish)
  unsigned* ready;
  unsigned seq;
  int publish, kbase;
  DeviceConfig cfg;
};
struct Empty {};

// (The body is a device function so that the SPLIT kernel can carry its own launch bounds: it needs the register cap of
//  two waves per SIMD, and a second __launch_bounds__ argument -- which cannot be left out conditionally -- makes the
//  compiler move a dynamically indexed private array of the rank-split shape into LDS: 8 KB more per workgroup and
//  that kernel 2.4 x slower at 8 trajectories.)
template <int MCAP, int NW, bool KSPLIT, bool SPLIT>
__device__ __forceinline__ void panels_mono(double* __restrict__ P, double* __restrict__ V,
                                            double* __restrict__ W, const double* __restrict__ mu_in,
                                            double* __restrict__ mu_out, const int* __restrict__ nact,
                                            const SolveOut* __restrict__ so,
                                            const double* __restrict__ fac, int ld, long pstride,
                                            const SplitArgs& sa) {
  static_assert(!SPLIT || (!KSPLIT && NW == 4), "SPLIT is the throughput shape");
  constexpr int CC = 3 + 2 * MCAP, KTP = ranks_for(MCAP), NT = 64 * NW;
  __shared__ __attribute__((aligned(16))) double sF[2][CC][KTOT];   // [0]: W[C[a]][k], [1]: V[k][C[a]]
  __shared__ SolveIter sIt[MCAP];
  __shared__ int sC[CC + 1];
  __shared__ double sPart[KSPLIT ? NW - 1 : 1][KSPLIT ? CC : 1][64];   // KSPLIT: partial gathers of waves 1..
  __shared__ std::conditional_t<SPLIT, SolveCore, Empty> sL;          // SPLIT: the solve's own LDS (factors: sF)
  __shared__ std::conditional_t<SPLIT, SolveHead, Empty> sH;          // SPLIT: the header fetched from the mailbox
  const int b = blockIdx.y;
  const int n = nact[b];
  const bool solver = SPLIT && blockIdx.x == 0;
  const bool early = SPLIT && !solver;                 // this workgroup gathers before the solve has finished
  // (SPLIT: what the solve of THIS launch writes is read through sa.out, which carries no read-only promise)
  const int w0 = (SPLIT ? (int)blockIdx.x - 1 : (int)blockIdx.x) * (KSPLIT ? 64 : NT);
  if (!solver && w0 >= n) return;
  // (SPLIT: a panel workgroup reads the header it fetched into LDS; the solve workgroup what it wrote itself)
  const SolveHead& o = [&]() -> const SolveHead& {
    if constexpr (SPLIT) return early ? static_cast<const SolveHead&>(sH) : static_cast<const SolveHead&>(sa.out[b]);
    else return so[b];
  }();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ld16 = ld >> 4;
  double* Pb = P + (long)b * pstride;
  double* Vb = V + (long)b * KTOT * ld;
  double* Wb = W + (long)b * KTOT * ld;
  const double* mu_in_b = mu_in + (long)b * ld;
  double* mu_out_b = mu_out + (long)b * ld;
  int kb, neff, m, c, cmaxv, Cl = 0;
  if constexpr (SPLIT) {
    const int neff_eff = min(n, max(sa.in[b].neff, sa.neff_floor[b]));
    if (solver) {
      if (b == 0 && tid < 8) sa.queue[tid * RS_QSTRIDE] = 0u;   // (see k_solve)
      solve_body(sL, FacPanel<CC>{sF}, Pb, Vb, Wb, sa.dacc_in + 4 * b, sa.dacc_out + 4 * b, mu_in_b, mu_out_b, sa.in[b],
                 sa.out[b], sa.out[b].it, sa.flags + b, sa.fac + (long)b * FACS, sa.cfg, ld, sa.kbase, true, neff_eff);
      {
        int ms = ((sa.in[b].flags & FLAG_UPDATE) && sa.cfg.enable_measurement_model) ? sa.in[b].m : 0;
        mailbox_publish(sa.out[b], sa.mbox + b, min(ms, MMAX), sa.ready + b, sa.seq, sa.publish);
      }
      return;
    }
    const StepIn& st = sa.in[b];                       // the gathered indices, as the solve forms them
    const int my_idx = (lane >= 3 && lane < CMAX) ? st.idx[(lane - 3) >> 1] : 0;
    int mm = ((st.flags & FLAG_UPDATE) && sa.cfg.enable_measurement_model) ? st.m : 0;
    mm = min(min(mm, MMAX), MCAP);
    m = mm;
    c = 3 + 2 * mm;
    Cl = (lane < 3) ? lane : (lane < c ? 3 + 2 * my_idx + ((lane - 3) & 1) : 0);
    cmaxv = Cl;
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) cmaxv = max(cmaxv, __shfl_xor(cmaxv, sh));
    kb = sa.kbase;
    neff = neff_eff;
  } else {
    kb = o.kbase;
    neff = o.neff;
    m = min(o.m, MCAP);
    c = o.c;
    cmaxv = o.cmax;
  }
  const int i0 = KSPLIT ? w0 : w0 + wave * 64;
  const int i = i0 + lane;
  const bool act = i < n;
  const int ii = act ? i : n - 1;                      // idle lanes shadow the last state index (no stores)
  const int kw = KSPLIT ? wave : 0;                    // KSPLIT: this wave takes every NW-th group of 8 ranks
  constexpr int KSTEP = KSPLIT ? 8 * NW : 8;

  if (early) {
    // the factors at C for this workgroup, straight from V and W: sF[0][a][k] = W[C[a]][k], sF[1][a][k] = V[k][C[a]]
    // (zero up to a whole group of 8 ranks); wave w takes the rows w, w + 4, ..., lane = rank (see stage_issue)
    if (tid < CC + 1) sC[tid] = Cl;
    if (w0 < neff && kb > 0) {
      const __amdgpu_buffer_rsrc_t rsV = rs_rsrc(Vb), rsW = rs_rsrc(Wb);
      constexpr int Q = (CC + 3) / 4;
      const int k8 = (kb + 7) & ~7;
      for (int kofs = 0; kofs < kb; kofs += 64) {
        const int k = kofs + lane, kc = min(k, kb - 1);
        const unsigned kW = (unsigned)(((kc >> 2) * ld16) * 64 + (kc & 3) * 16) * 8u;
        const unsigned kV = (unsigned)(kc * ld) * 8u;
        double wv[Q], vv[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const int a = wave + 4 * q;
          wv[q] = 0.0;
          vv[q] = 0.0;
          if (a < c) {                                 // (wave-uniform)
            const int row = __builtin_amdgcn_readlane(Cl, a);
            wv[q] = ldb8(rsW, kW, (unsigned)((row >> 4) * 64 + (row & 15)) * 8u);
            vv[q] = ldb8(rsV, kV, (unsigned)row * 8u);
          }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const int a = wave + 4 * q;
          if (a < CC && k < k8) {
            const bool live = a < c && k < kb;
            sF[0][a][k] = live ? wv[q] : 0.0;
            sF[1][a][k] = live ? vv[q] : 0.0;
          }
        }
      }
    }
  } else if (w0 < neff) {                              // (uniform) some wave of this workgroup replays the step
    if (kb > 0) {
      const double2* src0 = reinterpret_cast<const double2*>((SPLIT ? sa.fac : fac) + (long)b * FACS);
      const double2* src1 = src0 + CMAX * KTOT / 2;
      double2* dst0 = reinterpret_cast<double2*>(&sF[0][0][0]);
      double2* dst1 = reinterpret_cast<double2*>(&sF[1][0][0]);
      constexpr int CNT = CC * KTOT / 2, Q = (CNT + NT - 1) / NT;
      double2 t0[Q], t1[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int t = min(tid + q * NT, CNT - 1);
        t0[q] = src0[t];
        t1[q] = src1[t];
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int t = tid + q * NT;
        if (t < CNT) {
          dst0[t] = t0[q];
          dst1[t] = t1[q];
        }
      }
    }
    {
      const SolveOut& og = [&]() -> const SolveOut& {
        if constexpr (SPLIT) return sa.out[b];
        else return so[b];
      }();
      const double2* src = reinterpret_cast<const double2*>(og.it);
      double2* dst = reinterpret_cast<double2*>(sIt);
      constexpr int PER = (int)(sizeof(SolveIter) / 16), CNT = MCAP * PER, Q = (CNT + NT - 1) / NT;
      const int count = m * PER;
      double2 t0[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) t0[q] = src[min(tid + q * NT, CNT - 1)];
#pragma unroll
      for (int q = 0; q < Q; ++q)
        if (tid + q * NT < count) dst[tid + q * NT] = t0[q];
    }
    if (tid < CC + 1) sC[tid] = o.C[tid];
  }
  // the base entries of the gather do not depend on the staged data: issue them before the barrier
  double X[CC];
  if (i0 < neff && i0 < n && kw == 0) {
#pragma unroll
    for (int a = 0; a < CC; ++a) {
      const int row = early ? __builtin_amdgcn_readlane(Cl, a) : o.C[a];
#ifdef PANELS_SKIP_COLG                                 /* diagnostic build: no column-direction gathers */
      X[a] = Pb[p_index(ld, min(row, i0), max(row, ii))];
#else
      X[a] = Pb[p_index(ld, min(row, ii), max(row, ii))];
#endif
    }
  } else {
#pragma unroll
    for (int a = 0; a < CC; ++a) X[a] = 0.0;
  }
  __syncthreads();
  if (i0 >= n) return;
  if (i0 >= neff) {
    // beyond the active bound the rows and columns of P are exactly zero off the diagonal: this step's
    // ranks are zero there and the mean is carried over
    if (act && kw == 0) {
      for (int k = kb; k < ((kb + KTP + 3) & ~3); ++k) {
        Vb[(long)k * ld + i] = 0.0;
        Wb[wm_index(ld16, k, i)] = 0.0;
      }
      mu_out_b[i] = mu_in_b[i];
    }
    return;
  }

  // ---- pending ranks of the gather ----
  // The loop exists in two forms, with and without the W[i,:] loads (needed only where some gathered index lies
  // beyond this wave's first state index).  Inside a form no load sits under a branch -- the group of 8 ranks
  // after the last one is fetched too (rank rows up to KTOT exist) and masked -- so the compiler can count the
  // outstanding loads and the next group really is in flight under this group's FMAs.
  const double* vlane = Vb + ii;
  const double* wlane = Wb + (long)(ii >> 4) * 64 + (ii & 15);   // wm_index = rank part + lane part
  auto gather_pending = [&](auto need_w_tag) {
  constexpr bool NEEDW = decltype(need_w_tag)::value;
  double v[8], w[8];
  // (the group after the last one is "fetched" too so that no load sits under a branch, but it is pointed at the
  //  last valid group again: rows that were just read, not rank slots nobody needs -- a V row costs a trip to HBM)
  const int k_last = ((kb - 1) >> 3) << 3;
  auto load_vw = [&](int k0, double (&vv)[8], double (&ww)[8]) {
    const int kc = min(k0, k_last);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double x = vlane[(long)(kc + u) * ld];
      vv[u] = (k0 + u < kb) ? x : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (NEEDW) {
        const double x = wlane[((long)((kc + u) >> 2) * ld16) * 64 + ((kc + u) & 3) * 16];
        ww[u] = (k0 + u < kb) ? x : 0.0;
      } else {
        ww[u] = 0.0;
      }
    }
  };
  load_vw(8 * kw, v, w);
  for (int k0 = 8 * kw; k0 < kb; k0 += KSTEP) {
    double vn[8], wn[8];
    load_vw(k0 + KSTEP, vn, wn);
    // four gathered rows at a time: their 16 coefficient reads are in flight together, and each row's eight
    // FMAs run as two independent chains (the branch per row is wave-uniform)
#pragma unroll
    for (int g = 0; g < CC; g += 4) {
      double cf[4][8];                                 // (plain doubles: arrays of double2 end up in scratch)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int a = min(g + q, CC - 1);              // (rows c.. of a short step hold stale, unused data)
        const double2* src = reinterpret_cast<const double2*>(&sF[sC[a] <= i0 ? 0 : 1][a][k0]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double2 t = src[u];
          cf[q][2 * u] = t.x;
          cf[q][2 * u + 1] = t.y;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int a = g + q;
        if (a < CC) {
          const int row = sC[a];
          double s0 = X[a], s1 = 0.0;
          if (row <= i0) {                             // stored as (C[a], i) for the whole wave
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              s0 = fma(cf[q][2 * u], v[2 * u], s0);
              s1 = fma(cf[q][2 * u + 1], v[2 * u + 1], s1);
            }
          } else if (row > i0 + 63) {                  // mirrored for the whole wave
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              s0 = fma(w[2 * u], cf[q][2 * u], s0);
              s1 = fma(w[2 * u + 1], cf[q][2 * u + 1], s1);
            }
          } else {                                     // the wave straddles C[a]: both forms, chosen per lane
            const double2* wa = reinterpret_cast<const double2*>(&sF[0][a][k0]);
            const bool up = row <= ii;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const double2 fw2 = wa[u];
              s0 = fma(up ? fw2.x : w[2 * u], up ? v[2 * u] : cf[q][2 * u], s0);
              s1 = fma(up ? fw2.y : w[2 * u + 1], up ? v[2 * u + 1] : cf[q][2 * u + 1], s1);
            }
          }
          X[a] = s0 + s1;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      v[u] = vn[u];
      w[u] = wn[u];
    }
  }
  };
#ifndef PANELS_SKIP_PEND                                  /* diagnostic build: no pending-rank gather */
  if (kb > 0) {                                        // (uniform) right after a covariance pass nothing is pending
    if (cmaxv > i0) gather_pending(std::true_type{});
    else gather_pending(std::false_type{});
  }
#endif
  if constexpr (SPLIT) {
    if (early) {
      // Wait for the solve of this trajectory (every wave for itself: some waves of the last block have left), then
      // fetch its header and records from the mailbox -- every wave writes the same values into sH and sIt.
      const bool ok = mailbox_fetch(sa.mbox + b, m, sa.ready + b, sa.seq, SPLIT_SPIN_LIMIT, &sH, sIt);
      if (!ok) {                                       // (wave-uniform) the solve never reported: this wave has written
        if (lane == 0) atomicOr(sa.flags + b, EKF_FLAG_INTERNAL);   // nothing yet and writes nothing -- P_base, V, W and the
        return;                                        // mean stay as they were; the host turns the flag into EKF_ERR_STATE
      }
    }
  }
  if (KSPLIT) {                                        // waves 1.. hand their partial sums to wave 0 and leave
    if (kw > 0) {
#pragma unroll
      for (int a = 0; a < CC; ++a) sPart[kw - 1][a][lane] = X[a];
    }
    __syncthreads();
    if (kw > 0) return;
#pragma unroll
    for (int a = 0; a < CC; ++a) {
      double t = X[a];
#pragma unroll
      for (int q = 0; q < NW - 1; ++q) t += sPart[q][a][lane];
      X[a] = t;
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
    if (ii == a) X[a] += o.dacc_old[a];                // pending pose-block noise on the diagonal

  // ---- predict:  P' = G_F P G_F^T + F^T R F  (src/replay_no_ros.py:430) on the panel ----
  // Of the stored triangle it changes rows 0 and 1 only: thread i adds its two entries to P_base directly
  // (the pending ranks are unaffected), so the prediction costs no rank of the covariance pass.
  const double g0 = o.g[0], g1 = o.g[1];
  const double gj = (ii == 0) ? g0 : ((ii == 1) ? g1 : 0.0);
  double d0, d1;                                       // P'(0,i) - P(0,i),  P'(1,i) - P(1,i)
  if (i0 == 0) {
    if (lane < 2) {                                    // rows 0,1 of P_base are being rewritten by the other columns:
#pragma unroll
      for (int a = 0; a < CC; ++a) X[a] = o.prow[lane][a];   // state indices 0,1 take k_solve's gather
    }
    const double p22 = __shfl(X[2], 2);
    double col2[CC];                                   // X[:,2] after the row ops, for the column ops of lanes 0,1
#pragma unroll
    for (int a = 0; a < CC; ++a) col2[a] = __shfl(X[a], 2);
    col2[0] += g0 * p22;
    col2[1] += g1 * p22;
    d0 = g0 * X[2];                                    // row ops on rows 0,1
    d1 = g1 * X[2];
    if (ii < 2) {
      d0 += gj * col2[0];
      d1 += gj * col2[1];
#pragma unroll
      for (int a = 2; a < CC; ++a) X[a] += gj * col2[a];
    }
  } else {
    d0 = g0 * X[2];
    d1 = g1 * X[2];
  }
  X[0] += d0;
  X[1] += d1;
  if (act) {
    double* p0 = Pb + p_col(ld, i);                    // entry (0, i)
    *p0 += d0;
    if (i >= 1) p0[p_lds(ld)] += d1;                   // entry (1, i); (1, 0) lies below the diagonal
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
    if (a == ii) X[a] += o.rd[a];

  // ---- sequential landmark updates ----
  double dm = 0.0;
#pragma unroll
  for (int it = 0; it < MCAP; ++it) {
    const int kr = kb + 2 * it;
    if (it < m) {
      const SolveIter& I = sIt[it];
      const int a0 = 3 + 2 * it;
      double2 hk[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) hk[k] = *reinterpret_cast<const double2*>(I.h5t[k]);
      const double2 s01 = *reinterpret_cast<const double2*>(&I.si[0]);
      const double2 s23 = *reinterpret_cast<const double2*>(&I.si[2]);
      const double2 yy = *reinterpret_cast<const double2*>(I.y);
      double e0 = hk[0].x * X[0], e1 = hk[0].y * X[0];   // (H_j P_j)[:, i] = h5 . x[sel]
#pragma unroll
      for (int k = 1; k < 5; ++k) {
        const double xv = (k < 3) ? X[k] : X[a0 + (k - 3)];
        e0 = fma(hk[k].x, xv, e0);
        e1 = fma(hk[k].y, xv, e1);
      }
      const double f0 = e0 * s01.x + e1 * s23.x;       // K_j[i,:] = (H_j P_j)[:, i]^T S^-1  (P symmetric)
      const double f1 = e0 * s01.y + e1 * s23.y;
      dm += f0 * yy.x + f1 * yy.y;
      if (act) {
        Vb[(long)kr * ld + i] = e0;
        Vb[(long)(kr + 1) * ld + i] = e1;
        Wb[wm_index(ld16, kr, i)] = -f0;
        Wb[wm_index(ld16, kr + 1, i)] = -f1;
      }
      if (it + 1 < m) {                                // x[a] -= K_j[C[a],:] . (H_j P_j)[:, i]
        // only the rows a later landmark still reads: the pose rows and the rows of landmarks it+1..
#pragma unroll
        for (int a = 0; a < CC; ++a) {
          if (a < 3 || a >= a0 + 2) {
            const double2 kc = *reinterpret_cast<const double2*>(I.kc[a]);
            X[a] = fma(-kc.x, e0, X[a]);
            X[a] = fma(-kc.y, e1, X[a]);
          }
        }
      }
    } else if (act) {
      Vb[(long)kr * ld + i] = 0.0;
      Vb[(long)(kr + 1) * ld + i] = 0.0;
      Wb[wm_index(ld16, kr, i)] = 0.0;
      Wb[wm_index(ld16, kr + 1, i)] = 0.0;
    }
  }
  if (act) {
    for (int k = kb + KTP; k < ((kb + KTP + 3) & ~3); ++k) {   // k-tile pad
      Vb[(long)k * ld + i] = 0.0;
      Wb[wm_index(ld16, k, i)] = 0.0;
    }
    bool inC = false;
    for (int a = 0; a < c; ++a) inC |= (sC[a] == i);
    if (!inC) mu_out_b[i] = mu_in_b[i] + dm;           // k_solve wrote the entries in C
  }
}

template <int MCAP, int NW, bool KSPLIT>
__global__ __launch_bounds__(64 * NW) void k_panels(double* __restrict__ P, double* __restrict__ V,
                                                    double* __restrict__ W, const double* __restrict__ mu_in,
                                                    double* __restrict__ mu_out, const int* __restrict__ nact,
                                                    const SolveOut* __restrict__ so,
                                                    const double* __restrict__ fac, int ld, long pstride) {
  panels_mono<MCAP, NW, KSPLIT, false>(P, V, W, mu_in, mu_out, nact, so, fac, ld, pstride, SplitArgs{});
}
template <int MCAP>
__global__ __launch_bounds__(256, 2) void k_panels_split(double* __restrict__ P, double* __restrict__ V,
                                                         double* __restrict__ W, const double* __restrict__ mu_in,
                                                         double* __restrict__ mu_out, const int* __restrict__ nact,
                                                         int ld, long pstride, SplitArgs sa) {
  panels_mono<MCAP, 4, false, true>(P, V, W, mu_in, mu_out, nact, nullptr, nullptr, ld, pstride, sa);
}

// ---------------------------------------------------------------------------------------------
// k_step_split: the whole step in ONE launch for launches of few workgroups (the latency regime: at most 512 waves
// of state indices, where k_panels runs in its rank-split shape).  Workgroup 0 of a trajectory is the sequential solve
// (exactly k_solve); the others are panel workgroups of 64 state indices, which do NOT wait for it to start: nothing
// in the gather of the panel depends on the solve -- the gathered indices C come from the step's inputs, the factors
// at C are staged by every panel workgroup for itself (scattered loads the idle CUs have time for) -- so base entries
// and pending ranks are gathered while the chain runs.  Only then does a panel workgroup wait for its trajectory's
// solve (release / acquire at device scope on a per-trajectory step counter: the solve's records travel through
// global memory; with this few workgroups the two L2 flushes are cheaper than the mailbox of k_panels_split: 19.0 us
// per step against 21.1 us), replay the step on the gathered panel and write its ranks.  Every write of a panel workgroup
// happens after that wait, so the solve never sees a half-written step.  A workgroup only ever waits for one with a
// smaller linear index (dispatched before it), the wait is bounded, and a timeout raises EKF_FLAG_INTERNAL.
// ---------------------------------------------------------------------------------------------

template <int MCAP>
__global__ __launch_bounds__(256) void k_step_split(double* __restrict__ P, double* __restrict__ V,
                                                    double* __restrict__ W, const double* __restrict__ dacc_in,
                                                    double* __restrict__ dacc_out, const double* __restrict__ mu_in,
                                                    double* __restrict__ mu_out, const int* __restrict__ nact,
                                                    const StepIn* __restrict__ in, SolveOut* out,
                                                    unsigned* __restrict__ flags, double* __restrict__ fac,
                                                    const int* __restrict__ neff_floor, unsigned* __restrict__ queue,
                                                    unsigned* __restrict__ ready, unsigned seq, int publish,
                                                    DeviceConfig cfg, int ld, long pstride, int kbase) {
  constexpr int CC = 3 + 2 * MCAP;
  struct PanelSide {
    PanelLds<MCAP, 4, true> S;
    PanelPartLds<MCAP, 3> SP;
  };
  __shared__ union {
    SolveLds L;
    PanelSide Pn;
  } U;
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = nact[b];
  const int neff_eff = min(n, max(in[b].neff, neff_floor[b]));
  if (blockIdx.x == 0) {                               // ---- the solve of trajectory b ----
    if (b == 0 && tid < 8) queue[tid * RS_QSTRIDE] = 0u;   // (see k_solve)
    solve_body(U.L, U.L.fac_view(), P + (long)b * pstride, V + (long)b * KTOT * ld, W + (long)b * KTOT * ld, dacc_in + 4 * b,
               dacc_out + 4 * b, mu_in + (long)b * ld, mu_out + (long)b * ld, in[b], out[b], out[b].it, flags + b,
               fac + (long)b * FACS, cfg, ld, kbase, true, neff_eff);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // this wave's stores are visible device-wide ...
    __syncthreads();                                   // ... every wave's are ...
    // ... then the word (publish = 0: a test of the panels' bounded wait -- they must time out, not hang)
    if (tid == 0 && publish) __hip_atomic_store(ready + b, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // ---- a panel workgroup: 64 state indices, four waves splitting the pending ranks ----
  const int w0 = (blockIdx.x - 1) * 64;
  if (w0 >= n) return;
  auto& S = U.Pn.S;
  auto& SP = U.Pn.SP;
  const StepIn& s = in[b];
  // the gathered indices, as the solve forms them (lane a holds C[a], 0 beyond c)
  const int my_idx = (lane >= 3 && lane < CMAX) ? s.idx[(lane - 3) >> 1] : 0;
  int m = ((s.flags & FLAG_UPDATE) && cfg.enable_measurement_model) ? s.m : 0;
  m = min(min(m, MMAX), MCAP);
  const int c = 3 + 2 * m;
  const int Cl = (lane < 3) ? lane : (lane < c ? 3 + 2 * my_idx + ((lane - 3) & 1) : 0);
  int cmax = Cl;
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) cmax = max(cmax, __shfl_xor(cmax, sh));
  const int kb = kbase;
  const PanelIdx t = panel_idx(P, V, W, mu_in, mu_out, ld, pstride, b, n, w0);
  if (w0 >= neff_eff) {                                // beyond the active bound: nothing to wait for
    if (wave == 0) panel_beyond_bound<MCAP>(t, kb);
    return;
  }
  if (tid < CC + 1) S.sC[tid] = Cl;
  // the factors at C for this workgroup: sF[0][a][k] = W[C[a]][k], sF[1][a][k] = V[k][C[a]] (zero up to a whole group
  // of 8 ranks); wave w takes the rows w, w + 4, ..., lane = rank (see stage_issue for the addressing)
  if (kb > 0) {
    const __amdgpu_buffer_rsrc_t rsV = rs_rsrc(t.Vb), rsW = rs_rsrc(t.Wb);
    constexpr int Q = (CC + 3) / 4;
    const int k8 = (kb + 7) & ~7;
    for (int kofs = 0; kofs < kb; kofs += 64) {
      const int k = kofs + lane, kc = min(k, kb - 1);
      const unsigned kW = (unsigned)(((kc >> 2) * (ld >> 4)) * 64 + (kc & 3) * 16) * 8u;
      const unsigned kV = (unsigned)(kc * ld) * 8u;
      double wv[Q], vv[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int a = wave + 4 * q;
        wv[q] = 0.0;
        vv[q] = 0.0;
        if (a < c) {                                   // (wave-uniform)
          const int row = __builtin_amdgcn_readlane(Cl, a);
          wv[q] = ldb8(rsW, kW, (unsigned)((row >> 4) * 64 + (row & 15)) * 8u);
          vv[q] = ldb8(rsV, kV, (unsigned)row * 8u);
        }
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int a = wave + 4 * q;
        if (a < CC && k < k8) {
          const bool live = a < c && k < kb;
          S.sF[0][a][k] = live ? wv[q] : 0.0;
          S.sF[1][a][k] = live ? vv[q] : 0.0;
        }
      }
    }
  }
  double X[CC];
  if (wave == 0) {
    int Crow[CC];
#pragma unroll
    for (int a = 0; a < CC; ++a) Crow[a] = __builtin_amdgcn_readlane(Cl, a);
    panel_base_gather<CC>(t, Crow, X);
  } else {
#pragma unroll
    for (int a = 0; a < CC; ++a) X[a] = 0.0;
  }
  __syncthreads();
  if (kb > 0) {
    const double* vlane = t.Vb + t.ii;
    const double* wlane = t.Wb + (long)(t.ii >> 4) * 64 + (t.ii & 15);   // wm_index = rank part + lane part
    auto gather_pending = [&](auto need_w_tag) {
      constexpr bool NEEDW = decltype(need_w_tag)::value;
      double v[8], w[8];
      pg_load<NEEDW>(vlane, wlane, ld, t.ld16, kb, 8 * wave, v, w);
      for (int k0 = 8 * wave; k0 < kb; k0 += 32) {
        double vn[8], wn[8];
        pg_load<NEEDW>(vlane, wlane, ld, t.ld16, kb, k0 + 32, vn, wn);
        pg_accumulate<CC>(X, v, w, S.sF, S.sC, k0, w0, t.ii);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v[u] = vn[u];
          w[u] = wn[u];
        }
      }
    };
    if (cmax > w0) gather_pending(std::true_type{});
    else gather_pending(std::false_type{});
    if (wave > 0) {
#pragma unroll
      for (int a = 0; a < CC; ++a) SP.sPart[wave - 1][a][lane] = X[a];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int a = 0; a < CC; ++a) {
      double x = X[a];
#pragma unroll
      for (int q = 0; q < 3; ++q) x += SP.sPart[q][a][lane];
      X[a] = x;
    }
  } else if (wave > 0) {
    return;
  }
  // ---- wave 0: wait for the solve of this trajectory, fetch its records, replay ----
  {
    unsigned got = 0;
    for (int spin = 0; spin < SPLIT_SPIN_LIMIT; ++spin) {
      got = __hip_atomic_load(ready + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (got == seq) break;
      __builtin_amdgcn_s_sleep(4);
    }
    if (got != seq) {                                  // (wave-uniform) the solve never reported: nothing has been written by
      if (lane == 0) atomicOr(flags + b, EKF_FLAG_INTERNAL);   // this workgroup and nothing will be -- P_base, V, W and the
      return;                                          // mean stay as they were; the host turns the flag into EKF_ERR_STATE
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  const SolveOut& o = out[b];
  {
    const double2* src = reinterpret_cast<const double2*>(o.it);
    double2* dst = reinterpret_cast<double2*>(S.sIt);
    constexpr int PER = (int)(sizeof(SolveIter) / 16), CNT = MCAP * PER, Q = (CNT + 63) / 64;
    const int count = m * PER;
    double2 t0[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) t0[q] = src[min(lane + q * 64, CNT - 1)];
#pragma unroll
    for (int q = 0; q < Q; ++q)
      if (lane + q * 64 < count) dst[lane + q * 64] = t0[q];
  }
  WAVE_LDS_SYNC();
  panels_finish<MCAP>(t, o, S.sIt, S.sC, X);
}

// ---------------------------------------------------------------------------------------------
// k_flush: P_base[i][j] += sum_{k < 4 nkt} W[i][k] V[k][j] + dacc on the pose diagonal, in place, on the
// stored (upper) triangle: one read + one write of it (8 n(n+1) bytes per trajectory) for ALL pending
// steps, with the rank-K product on the fp64 matrix cores.
// A wave owns a strip of 64 columns; the V strip lives in registers as MFMA B fragments (nkt x 4
// doubles per lane) for the whole row block.  Per 16-row tile:
//   global -> registers   8 x global_load_dwordx4: every instruction is two full 512-B row segments
//                         (the tile after this one is prefetched while this one computes)
//   registers -> LDS      row-major image, wave-private (no workgroup barrier anywhere)
//   LDS -> accumulators   16 x ds_read_b64 in the fp64 C/D layout (col = lane&15, row = lane>>4 + 4 reg);
//                         the row stride of 80 doubles puts the two rows of a 32-lane group on
//                         disjoint bank halves
//   W tile                A fragments, one lane-linear 512-B load per k-tile (wm_index layout), L2-resident
//   4 x nkt v_mfma_f64_16x16x4_f64, then the same path back (LDS transpose, 1 KiB stores).
// Going through LDS keeps every HBM access 16 B per lane; loading the C/D layout straight from global
// memory (8 B per lane, 128-B segments) ran at 0.6x the streaming rate.
// NT: nontemporal loads/stores for working sets beyond the 256 MiB Infinity Cache.
// Bound: HBM up to ~48 pending ranks (0.80 ms for 32 x N=2000: 5.1 TB/s); beyond, the MFMA phase of a
// wave no longer hides under its own memory wait (0.885 ms at 64 ranks, 0.99 ms at 80; the matrix pipe is
// 55 % busy there -- v_mfma_f64_16x16x4 sustains 77.9 TFLOP/s on this part, profiles/mfma_probe.txt).
// ---------------------------------------------------------------------------------------------
template <bool NT>
__device__ __forceinline__ double2 ld2(const double* a) {
  double2 r;
  if (NT) {
    r.x = __builtin_nontemporal_load(a);
    r.y = __builtin_nontemporal_load(a + 1);
  } else {
    r = *reinterpret_cast<const double2*>(a);
  }
  return r;
}
template <bool NT>
__device__ __forceinline__ void st2(double* a, double2 v) {
  if (NT) {
    __builtin_nontemporal_store(v.x, a);
    __builtin_nontemporal_store(v.y, a + 1);
  } else {
    *reinterpret_cast<double2*>(a) = v;
  }
}

constexpr int FTS = 80;                 // LDS row stride of a wave's 16 x 64 tile image (doubles)

template <int NKTM, int NKL, bool NT>
__global__ __launch_bounds__(256, 2) void k_flush(double* __restrict__ P, const double* __restrict__ V,
                                                  const double* __restrict__ W,
                                                  const double* __restrict__ dacc,
                                                  const int* __restrict__ nact,
                                                  const SolveOut* __restrict__ so, int ld, long pstride,
                                                  int nkt, int rows_per_block, int gx) {
  __shared__ double tiles[4][16 * FTS];
  // k-tiles NKTM.. of the V strip live in LDS (B fragments, lane-linear): the register file holds 15-16 k-tiles
  // at 2 waves/SIMD, the remaining LDS holds 5 more per wave
  __shared__ double vlds[NKL > 0 ? 4 : 1][NKL > 0 ? NKL * 4 * 64 : 1];
  const int b = blockIdx.z;
  const int n = min(nact[b], so[b].neff);              // rows/cols beyond the active bound are untouched
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  // Only workgroups that touch the upper triangle are launched: blockIdx.x enumerates, row block by row
  // block, the column groups bx >= by * rows_per_block / 256.  XCD-aware order (speed only): workgroups are
  // dealt round-robin over the 8 XCDs by linear id, so consecutive ids are remapped to give every XCD one
  // contiguous, equally long piece of that enumeration: the W rows of a row block (and the V strips) are
  // fetched into one L2 instead of eight, and the triangle is split evenly.
  int bx, by = 0;
  {
    const int total = gridDim.x;
    const int lin = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = lin & 7, slot = lin >> 3;
    int rem = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    for (;; ++by) {
      const int cnt = gx - (by * rows_per_block) / 256;
      if (rem < cnt || cnt <= 0) break;
      rem -= cnt;
    }
    bx = (by * rows_per_block) / 256 + rem;
  }
  const int j0 = (bx * 4 + wave) * 64;
  const int i_begin = by * rows_per_block;
  const int i_lim = min(n, j0 + 64);                   // only the upper triangle is stored: tiles strictly below
  if (j0 >= n || i_begin >= i_lim) return;             // the diagonal are skipped (a diagonal tile computes a few
  const int i_end = min(i_lim, i_begin + rows_per_block);   // lower entries nobody reads)
  const int li = lane & 15, lq = lane >> 4;            // C/D layout coordinates
  const int rr = lane >> 5, rc = (lane & 31) * 2;      // row-major image: 2 rows per instruction
  const int ld16 = ld >> 4;
  double* Pb = P + (long)b * pstride;
  const double* Vb = V + (long)b * KTOT * ld;
  const double* Wb = W + (long)b * KTOT * ld;
  double* T = tiles[wave];
  int nct = (n - j0 + 15) >> 4;                        // column tiles of this strip that start below n
  if (nct > 4) nct = 4;

  double vf[NKTM][4];
#pragma unroll
  for (int t = 0; t < NKTM; ++t)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
      vf[t][ct] = (t < nkt && ct < nct) ? Vb[(long)(4 * t + lq) * ld + j0 + ct * 16 + li] : 0.0;
  double* VL = vlds[NKL > 0 ? wave : 0];
  if (NKL > 0) {
    double tmp[NKL > 0 ? NKL * 4 : 1];                 // all loads first: a store to LDS after each load serialises them
#pragma unroll
    for (int t = 0; t < NKL; ++t)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        tmp[t * 4 + ct] = Vb[(long)(4 * min(NKTM + t, nkt - 1) + lq) * ld + j0 + ct * 16 + li];
#pragma unroll
    for (int t = 0; t < NKL; ++t)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) VL[(t * 4 + ct) * 64 + lane] = (NKTM + t < nkt && ct < nct) ? tmp[t * 4 + ct] : 0.0;
    WAVE_SYNC();
  }

  // Every global access of the row loop is unconditional (rows and columns up to the padded size ld exist, their
  // W rows / V columns are zero, so padding is read and written back unchanged): with no branch around a load or a
  // store the compiler can count them, and waits for the tile it needs instead of draining the queue
  // (s_waitcnt vmcnt(0) would expose the previous tile's store latency and the W round trip every tile).
  double2 g[8];                                        // row-major registers of the tile in flight
  const int lds = p_lds(ld);                           // row stride of P (ekf_device.h: column panels beyond ld = 4096)
  double* Pj = Pb + p_col(ld, j0 + rc);                // this lane's two columns of row 0: the strip lies inside one panel
  auto gload = [&](int i0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) g[q] = ld2<NT>(Pj + (long)(i0 + 2 * q + rr) * lds);
  };
  gload(i_begin);
  const int i_last = i_begin + ((i_end - 1 - i_begin) & ~15);
  // W fragments: one batch per tile, issued as soon as the previous tile's MFMAs have issued (they free the
  // registers), i.e. a tile's LDS phases and stores ahead of their first use: an L2 round trip of W no longer
  // sits between the accumulator read and the first MFMA.  Loads complete in order; this batch sits between the
  // P prefetch of the next tile (older) and the one after it (younger), so waiting for it leaves the younger
  // prefetch and the stores in flight.
  // (Below 12 k-tiles the MFMA phase is too short to matter: the batch is fetched at the top of the tile.  The
  // 20-k-tile form keeps 15 k-tiles of the V strip in registers and 5 in LDS -- 80 KB of LDS per workgroup, two
  // workgroups per CU -- so that this second live set of fragments fits without spilling.)
  constexpr int NKT_ALL = NKTM + NKL;
  constexpr bool EARLYW = NKTM >= 12;
  double wf[NKT_ALL];
  auto wload = [&](int i0) {
    const double* wsrc = Wb + (long)(i0 >> 4) * 64 + lane;
#pragma unroll
    for (int t = 0; t < NKT_ALL; ++t) wf[t] = wsrc[(long)min(t, nkt - 1) * ld16 * 64];
  };
  if (EARLYW) wload(i_begin);
  for (int i0 = i_begin; i0 < i_end; i0 += 16) {
    const int i_next = min(i0 + 16, i_last);           // (the last tile re-reads itself)
    if (!EARLYW) wload(i0);
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<double2*>(&T[(2 * q + rr) * FTS + rc]) = g[q];
    WAVE_SYNC();
    gload(i_next);                                     // prefetch under the MFMAs
    double4_t acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ct][r] = T[(lq + 4 * r) * FTS + ct * 16 + li];
    WAVE_SYNC();
#pragma unroll
    for (int t = 0; t < NKTM; ++t)
      if (t < nkt) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[t], vf[t][ct], acc[ct], 0, 0, 0);
      }
#pragma unroll
    for (int t = 0; t < NKL; ++t)                      // k-tiles whose B fragments come from LDS
      if (NKTM + t < nkt) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[NKTM + t], VL[(t * 4 + ct) * 64 + lane], acc[ct], 0, 0, 0);
      }
    if (i0 == 0 && j0 == 0) {                          // pose-block noise accumulated since the last flush
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = lq + 4 * r;
        if (row < 3 && li == row) acc[0][r] += dacc[4 * b + row];
      }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(lq + 4 * r) * FTS + ct * 16 + li] = acc[ct][r];
    if (EARLYW) wload(i_next);                         // (the accumulators are dead: their registers are not needed twice)
    WAVE_SYNC();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = i0 + 2 * q + rr;
      const double2 o = *reinterpret_cast<const double2*>(&T[(2 * q + rr) * FTS + rc]);
      st2<NT>(Pj + (long)row * lds, o);
    }
    WAVE_SYNC();
  }
}

// ---------------------------------------------------------------------------------------------
// k_flush_rs: the same update, "row slab" form (the default for batches that stream through HBM).
// The roles of the two operands are swapped with respect to k_flush: a wave keeps the W fragments of ITS 16 rows
// in registers (A operands: 2 VGPRs per k-tile instead of the 8 a 64-column V strip costs) and walks along those
// rows one 16 x 64 tile at a time; the V strip of the current 64 columns (B operands) lives in LDS and is shared
// by the eight waves of the workgroup, which cover 128 consecutive rows.  With 40 instead of 160 operand registers
// there is room to software-pipeline the tiles inside one wave, so that no MFMA ever waits for memory or for the
// LDS transposes:
//   while tile t is in the matrix pipe (4 * NKT MFMAs on `acc`), the same wave
//     . sends the result of tile t-1 (parked in its LDS image) to HBM as 16-byte row segments,
//     . writes tile t+1 (already in registers, row-major) into the image, issues the HBM loads of tile t+2 into the
//       registers that frees (round 4: right there, not at the end of the tile), and reads the image back in the
//       C/D layout -> `accn`,
//     . stages its share of the V strip of step t+1 into the other half of the strip buffer;
//   at the tile boundary: one workgroup barrier (s_barrier only: nothing in flight is drained), acc -> image,
//   acc <- accn.
// Every global access inside the pipelined loop is unconditional (loads beyond the end of a slab are pointed at a
// small cache-resident dummy, by a select, not a branch) and a compiler builtin, stores included (an inline-asm store
// is invisible to the compiler's vmcnt bookkeeping: see stb16), so that the compiler can count what is in flight.
// The covariance's layout (ekf_device.h): row-major up to ld = 4096, column panels of 4096 doubles beyond -- a tile's
// sixteen row segments are 32 KB apart at every size (N = 8000: 5.0-5.4 -> 5.4-5.8 TB/s, profiles/r04_pass_layout.txt).
// Work distribution: persistent workgroups (one per CU: 144 KB of LDS), units = (trajectory, 128-row slab) handed
// out longest first from per-XCD queues (trajectory b belongs to queue b % 8: the slabs of a trajectory share V and
// run on one L2; an idle workgroup steals from the other queues).  A slab is walked from the right end of its rows
// towards the diagonal: the slabs of a trajectory start on the same V strip.
// The accumulation order per element is k_flush's (k-tiles ascending): the two kernels agree bit for bit.
// ---------------------------------------------------------------------------------------------

// Global accesses of k_flush_rs are buffer instructions: (128-bit resource in SGPRs: wave-uniform base) + (SGPR byte
// offset: the tile) + (ONE 32-bit VGPR: the lane's place inside the tile).  64-bit per-lane pointers for the eight
// rows of a tile would cost the registers the pipeline needs (the compiler does not form the saddr + voffset
// global instructions when the lane offset is defined outside the loop).  Offsets are unsigned 32-bit: a
// trajectory's P is at most 4 GB.
template <bool NT>
__device__ __forceinline__ double2 ldb16(__amdgpu_buffer_rsrc_t rs, unsigned lane_bytes, unsigned tile_bytes) {
  const uint4v_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_bytes, (int)tile_bytes, NT ? 2 : 0);
  double2 r;
  r.x = __builtin_bit_cast(double, uint2v_t{v.x, v.y});
  r.y = __builtin_bit_cast(double, uint2v_t{v.z, v.w});
  return r;
}
// The 16-byte store of k_flush_rs.  A 16-byte store reads its data registers in two passes; a VALU instruction issued
// right behind it that writes one of them can overtake the second pass (dwords 2-3 of the last lanes).  The compiler's
// hazard recogniser adds the wait states for global/flat stores and for buffer stores WITHOUT an SGPR offset only (it
// assumes the SGPR read covers it); on gfx950 the form used here (offen + SGPR soffset) was caught corrupting the low
// mantissa bits of the odd columns of a tile's last row pair (`buffer_store_dwordx4 v[76:79]` followed by
// `v_add_u32 v78`).  The guard (round 4 form): the store is the compiler's own builtin, and the two wait states are an
// asm statement that (a) clobbers memory -- so it stays behind the store -- and (b) takes the store's data registers as
// INPUTS -- so no instruction that overwrites them can be scheduled in front of it.  Whatever the scheduler puts between
// store and s_nop can therefore only add wait states; tools/isa_lint.py checks on the shipped machine code that no
// writer of a store's data registers follows it within two wait states.
// Round 3 emitted store + s_nop as ONE inline-asm block.  The compiler does not count an inline-asm store in vmcnt, so
// every wait for a LOAD issued before such stores also waited for the stores (in-order counter): the pass lost 6 - 18 us
// (N = 2000 x 32, same box, interleaved runs: 744 us with the builtin store against 751 - 762 us with the asm block --
// profiles/r04_pass_drift.txt; round 3 had measured "no difference" on single runs).
template <bool NT>
__device__ __forceinline__ void stb16(__amdgpu_buffer_rsrc_t rs, unsigned lane_bytes, unsigned tile_bytes, double2 v) {
#ifdef RS_SKIP_PMEM
  if (v.x == 1.2345e-300) /* never: keeps the value alive, drops the traffic */
#endif
  {
  const uint2v_t a = __builtin_bit_cast(uint2v_t, v.x), b = __builtin_bit_cast(uint2v_t, v.y);
  const uint4v_t d{a.x, a.y, b.x, b.y};
  __builtin_amdgcn_raw_buffer_store_b128(d, rs, (int)lane_bytes, (int)tile_bytes, NT ? 2 : 0);
  asm volatile("s_nop 1" ::"v"(d) : "memory");
  }
}
#define RS_CBAR() asm volatile("" ::: "memory")
// Diagnostic build (-DRS_STAMPS): wave 0 of every workgroup records s_memtime at fixed points of its first units into
// the words behind the queue heads (tools/rs_stamps.py reads them through ekf_debug_read).
#ifdef RS_STAMPS
#define RS_STAMP(k)                                                                             \
  do {                                                                                          \
    if (wave == 0 && unit_no < 3) {                                                             \
      unsigned long long t_;                                                                    \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
      if (lane == 0) stamp_buf[unit_no * 20 + (k)] = t_;                                        \
    }                                                                                           \
  } while (0)
#else
#define RS_STAMP(k) do { } while (0)
#endif

// (the work queues and the equal static shares of k_flush_rs: plain integer code for host and device, ekf_host_plan.h)
// PAN: the covariance lies in column panels (ld > 4096); otherwise the column offset of a strip is plain j * 8 -- the
// panel arithmetic (a shift, a multiplication and a mask per tile, all scalar) is compiled out for the sizes that do not
// need it: it cost the N = 2000 x 32 pass 5 - 10 us (profiles/r04_pass_drift.txt).
template <int NKT, bool NT, bool PAN, bool WV>
__global__ __launch_bounds__(512, 2) void k_flush_rs(double* __restrict__ P, const double* __restrict__ V,
                                                     const double* __restrict__ W,
                                                     const double* __restrict__ dacc,
                                                     const int* __restrict__ nact,
                                                     const SolveOut* __restrict__ so, int ld, long pstride,
                                                     int nkt, int batch, int nrb, int nch, int cs_arg, int mode,
                                                     unsigned* __restrict__ queue, const int* __restrict__ shares,
                                                     const CadOut* __restrict__ wv) {
  constexpr int RPW = NKT / 2;                         // ranks of a V strip each of the 8 waves stages
  __shared__ __attribute__((aligned(16))) double vbuf[2][NKT * 256];   // V strip as B fragments: [k-tile][col tile][lane]
  __shared__ __attribute__((aligned(16))) double img[8][16 * 64];      // per wave: 16 x 64 tile image (swizzled)
  __shared__ int s_unit;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lq = lane >> 4;            // C/D layout coordinates
  const int rr = lane >> 5, rc = (lane & 31) * 2;      // row-major image: 2 rows per instruction
  const int ld16 = ld >> 4;
  const int grp = blockIdx.x & 7;                      // workgroups with equal blockIdx % 8 share an XCD (speed only)
  double* T = img[wave];
  // Image of a tile: element (r, c) at r * 64 + (c ^ ((r & 1) << 4)): both the row-major 16-byte accesses and the
  // 8-byte accesses in the C/D layout (two rows of opposite parity per 32 lanes) are bank-conflict free at 8 KB.
  const int rm_base = rr * 64 + (rc ^ (rr << 4));      // + 128 q  for rows 2q + rr
  // C/D layout, element (lq + 4 reg, ct * 16 + li):  (lq + 4 reg) * 64 + ((ct * 16 + li) ^ ((lq & 1) << 4)), written as
  // one of two lane bases plus a compile-time offset (the swizzle swaps column tiles 2u, 2u + 1 in odd rows)
  const int cd_even = lq * 64 + li + 16 * (lq & 1), cd_odd = lq * 64 + li + 16 * (1 - (lq & 1));
  auto cd_index = [&](int ct, int reg) { return ((ct & 1) ? cd_odd : cd_even) + 16 * (ct & ~1) + 256 * reg; };
  // rank k = wave + 8 i of a strip: k-tile 2 i + (wave >> 2), row wave & 3 of the fragment
  const int stage_base = ((wave >> 2) * 4 + lq) * 64 + (wave & 3) * 16 + li;
  const unsigned pl8 = (unsigned)(PAN ? p_lds(ld) : ld) * 8u;   // bytes per row of P (ekf_device.h: column panels beyond ld = 4096)
  const unsigned loff = (unsigned)rr * pl8 + (unsigned)rc * 8u;   // lane part of a tile address, bytes (rows 2q + rr, columns rc, rc + 1)
  const unsigned lane8 = (unsigned)lane * 8u;

  // next unit: own queue first, then the others (thread 0 only; -1 = every queue is empty).  While the own queue has
  // units the head is bumped without looking first (one round trip instead of two); a queue found empty is only
  // looked at from then on (the heads are never bumped far beyond their counts).
#ifdef RS_STAMPS
  unsigned long long* stamp_buf = reinterpret_cast<unsigned long long*>(queue + 8 * RS_QSTRIDE) + blockIdx.x * 64;
  int unit_no = -1;
#endif
  bool own_empty = false;
  // a unit from this XCD's queue, else from the next non-empty one (see rs_queue_count / rs_queue_unit)
  auto pop = [&]() -> int {
    int found = -1;
    for (int a = 0; a < 8 && found < 0; ++a) {
      const int g2 = (grp + a) & 7;
      const int cnt = rs_queue_count(g2, batch, nrb, nch, mode);
      if (cnt == 0) continue;
      unsigned* head = queue + g2 * RS_QSTRIDE;
      if (a > 0 || own_empty) {
        if (__hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)cnt) continue;
      }
      const unsigned u = atomicAdd(head, 1u);
      if (u < (unsigned)cnt) found = rs_queue_unit(g2, (int)u, batch, nrb, nch, mode);
      else if (a == 0) own_empty = true;
    }
    return found;
  };
  int piece = 0;                                       // (mode 4) next piece of this workgroup's share
  for (;;) {
#ifdef RS_STAMPS
    ++unit_no;
#endif
    RS_STAMP(0);
    __syncthreads();                                   // every wave is done with the LDS of the previous unit
    RS_STAMP(1);
    int b, rb, u_start, u_count;                       // the unit: strips [u_start, u_start + u_count) of slab rb of trajectory b
    if (mode == 4) {                                   // equal static shares: this workgroup's own list, no queue
      if (piece >= RS_PIECES) return;
      const int* pc = shares + ((long)blockIdx.x * RS_PIECES + piece) * 4;
      ++piece;
      b = pc[0];
      rb = pc[1];
      u_start = pc[2];
      u_count = pc[3];
      if (u_count <= 0) return;                        // end of the share: all eight waves leave together
    } else {
      if (threadIdx.x == 0) s_unit = pop();
      __syncthreads();
      const int unit = __builtin_amdgcn_readfirstlane(s_unit);   // (an LDS load is a vector value to the compiler)
      if (unit < 0) return;                            // every queue is empty: all eight waves leave together
      const int code = unit & 1023;
      b = (unit >> 10) / nrb;
      rb = (unit >> 10) - b * nrb;
      u_count = code == 1023 ? (1 << 20) : cs_arg;     // strips per unit: the whole slab, or a chunk of it
      u_start = code == 1023 ? 0 : code * cs_arg;
    }
    RS_STAMP(2);
    const int n = min(nact[b], so[b].neff);            // rows/cols beyond the active bound are untouched
    const int i0 = rb * RS_ROWS;
    if (i0 >= n) continue;
    // The slab's strips run from the right end of its rows (the rightmost strip that starts below n) down to the one
    // holding the diagonal; a unit is `cs` consecutive strips of them (all of them when the batch alone fills the
    // chip: nch = 1).  Below, j_last / S / Sw describe THIS UNIT's strips.
    const int j_right = ((n - 1) >> 6) << 6;
    const int S_slab = ((j_right - i0) >> 6) + 1;
    if (u_start >= S_slab) continue;
    const int j_last = j_right - 64 * u_start;         // first (rightmost) strip of the unit
    const int S = min(u_count, S_slab - u_start);
    const bool has_diag = u_start + S == S_slab;       // the unit ends on the strip that holds the diagonal
    const int i0w = i0 + 16 * wave;
    // tiles of this wave: none if its rows lie beyond n; in the strip that holds the diagonal (columns i0..i0+63)
    // the rows of waves 4-7 lie strictly below it
    const int Sw = (i0w < n) ? ((wave < 4 || !has_diag) ? S : S - 1) : 0;
    double* Pb = P + (long)b * pstride;
    const double* Vb = V + (long)b * KTOT * ld;
    const double* Wb = W + (long)b * KTOT * ld;
    const __amdgpu_buffer_rsrc_t rsV = rs_rsrc(Vb), rsW = rs_rsrc(Wb);
    const unsigned ld8 = (unsigned)ld * 8u;            // bytes per rank row of V
    const unsigned prow = (unsigned)i0w * pl8;         // byte offset of this wave's first row inside the trajectory's P (panel 0)
    const double dd0 = dacc[4 * b], dd1 = dacc[4 * b + 1], dd2 = dacc[4 * b + 2];

    // Staging of a V strip: wave w moves the ranks w, w + 8, w + 16, ... (RPW of them) of the strip of step t (clamped
    // to an existing one) global -> registers -> B fragments in LDS.  With that interleaving the LDS address of rank
    // w + 8 i is ONE lane base plus the compile-time offset i * 4 KB (ranks k0 .. k0 + RPW - 1 would need a VGPR
    // address each, kept live across the whole tile loop).
    auto stage_load = [&](int t, auto i0_tag, auto cnt_tag, double* vs) {
      constexpr int I0 = decltype(i0_tag)::value, CNT = decltype(cnt_tag)::value;
      const int j = j_last - 64 * min(t, S - 1);
#pragma unroll
      for (int i = 0; i < CNT; ++i) vs[i] = ldb8(rsV, lane8, (unsigned)(wave + 8 * (I0 + i)) * ld8 + (unsigned)j * 8u);
    };
    auto stage_store = [&](double* dst, auto i0_tag, auto cnt_tag, const double* vs) {
      constexpr int I0 = decltype(i0_tag)::value, CNT = decltype(cnt_tag)::value;
      double* d0 = dst + stage_base;                   // B fragment slot of rank `wave`, this lane
#pragma unroll
      for (int i = 0; i < CNT; ++i) {                  // rank slots beyond the pending ones hold stale data: zero
        d0[(I0 + i) * 512] = (wave + 8 * (I0 + i) < 4 * nkt) ? vs[i] : 0.0;   // (masked here, not at the load: the select
      }                                                 //  would wait for the load where it is issued)
    };
    using I0_ = std::integral_constant<int, 0>;
    using IH_ = std::integral_constant<int, RPW / 2>;
    using IR_ = std::integral_constant<int, RPW>;
    auto wg_barrier = [&]() {                          // LDS stores landed, then s_barrier: global accesses stay in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      RS_CBAR();
    };

    if (Sw == 0) {                                     // (wave-uniform) staging and barriers only
      double vs[RPW];
      stage_load(0, I0_{}, IR_{}, vs);
      stage_store(vbuf[0], I0_{}, IR_{}, vs);
      wg_barrier();
      for (int t = 0; t < S - 1; ++t) {
        stage_load(t + 1, I0_{}, IR_{}, vs);
        stage_store(vbuf[(t + 1) & 1], I0_{}, IR_{}, vs);
        wg_barrier();
      }
      continue;
    }

    double2 g[8];                                      // row-major registers of the tile in flight
    // tile t of this wave; beyond its last tile the loads are pointed at 1 KB of the (read-only, cache resident)
    // SolveOut record instead: a select on the uniform base, offset and stride, no branch
    auto tile_off = [&](int t) -> unsigned {           // (a 64-column strip lies inside one column panel)
      const unsigned j = (unsigned)(j_last - 64 * t);
      return prow + (PAN ? p_col8(ld, j) : j * 8u);
    };
    auto gload = [&](int t) {
#ifdef RS_SKIP_PMEM                                     /* diagnostic build: the compute side alone */
      const bool ok = false;
#else
      const bool ok = t < Sw;
#endif
      const __amdgpu_buffer_rsrc_t rs = rs_rsrc(ok ? (const void*)Pb : (const void*)so);
      const unsigned off = ok ? tile_off(t) : 0u, ld8d = ok ? pl8 : 0u, loffd = ok ? loff : lane8 * 2u;
#pragma unroll
      for (int q = 0; q < 8; ++q) g[q] = ldb16<NT>(rs, loffd, off + (unsigned)q * 2u * ld8d);
    };
    // two accumulator sets: tile t accumulates in accs[t & 1] while tile t+1 is brought into accs[(t + 1) & 1]
    // (compile-time roles -- the tile loop is unrolled by two -- so that no register copies sit between the MFMAs)
    double4_t accs[2][4];
    double wf[NKT];
    // B fragments: two register sets with compile-time roles (k-tile kt multiplies out of set kt & 1 while set
    // (kt + 1) & 1 is being read; NKT is even, so set 0 is also where the last k-tile of a tile prefetches the first
    // fragments of the NEXT tile): the reads of the next k-tile are always in flight under this one's MFMAs
    double bf[2][4];

    RS_STAMP(3);
    // ---- prologue: W fragments, strip 0, tile 0 -> accs[0], tile 1 in flight ----
    {
      double vs[RPW];
      stage_load(0, I0_{}, IR_{}, vs);
      gload(0);
      if constexpr (!WV) {                             // (WV: an instantiation of its own -- the plain kernel's code is untouched)
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
          const double x = ldb8(rsW, lane8, ((unsigned)t * ld16 + (unsigned)(i0w >> 4)) * 512u);
          wf[t] = (t < nkt) ? x : 0.0;
        }
      } else {
        // ("w_from_v", round 6) The pending ranks are those of ONE fused cadence whose panel launch wrote V only: W is
        // redundant -- per landmark W[i][2s .. 2s+1] = -(V[2s][i], V[2s+1][i]) S_s^-1, with S_s^-1 in the cadence's records --
        // and half of what that launch wrote.  This lane's fragment entry is rank k = 4t + lq: landmark s = 2t + (lq >> 1),
        // column c = lq & 1 of S^-1; the same operations as the panel launch's f = fma(e0, S[0][c], e1 * S[1][c]), W = -f.
        // The pose's state indices (rows 0..2, the solve's: CadOut::posevw) keep their stored entries.
        const CadOut& oc = wv[b];
        const int nsl = min(oc.nslots, CAD_SLOTS), s0c = CAD_SLOTS - nsl;
        const __amdgpu_buffer_rsrc_t rsR = rs_rsrc(oc.rec);
        const unsigned vlane = (unsigned)(2 * (lq >> 1)) * ld8 + (unsigned)li * 8u;
        const unsigned cpart = (unsigned)(lq & 1) * 8u;
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
          const int sl = 2 * t + (lq >> 1), slot = min(s0c + sl, CAD_SLOTS - 1);
          const unsigned roff = (unsigned)cad_rec_off(slot) * 8u + cpart;
          const unsigned voff = ((unsigned)(4 * t) * (unsigned)ld + (unsigned)i0w) * 8u;
          const double e0 = ldb8(rsV, vlane, voff), e1 = ldb8(rsV, vlane + ld8, voff);
          const double sa = ldb8(rsR, roff, 80u), sb = ldb8(rsR, roff, 96u);
          double w = -__builtin_fma(e0, sa, e1 * sb);
          if (i0w == 0) {                              // (uniform) the wave that holds the pose rows
            const double x = ldb8(rsW, lane8, (unsigned)t * ld16 * 512u);
            w = li < 3 ? x : w;
          }
          wf[t] = (t < nkt && sl < nsl) ? w : 0.0;
        }
      }
      stage_store(vbuf[0], I0_{}, IR_{}, vs);
      RS_STAMP(4);
#pragma unroll
      for (int q = 0; q < 8; ++q) *reinterpret_cast<double2*>(&T[rm_base + 128 * q]) = g[q];
      RS_CBAR();
      RS_STAMP(5);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) accs[0][ct][r] = T[cd_index(ct, r)];
      RS_CBAR();
      gload(1);
      wg_barrier();
      RS_STAMP(6);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) bf[0][ct] = vbuf[0][ct * 64 + lane];
    }

    // ---- one tile: the MFMAs of tile t with everything else of tiles t-1, t+1, t+2 issued between them.
    // No instruction of the tile boundary sits between the last MFMA of tile t and the first of tile t+1: the
    // result of tile t stays in its accumulator set and goes to the image at the start of tile t+1 (which
    // accumulates in the other set); the workgroup barrier that publishes the next V strip comes one k-tile
    // early (every B fragment of this tile has been read by then) and the last k-tile prefetches the first
    // fragments of the next tile.
    auto body = [&](auto first_tag, auto stage_tag, auto par_tag, int t) {
      constexpr bool FIRST = decltype(first_tag)::value, STAGE = decltype(stage_tag)::value;
      constexpr int PAR = decltype(par_tag)::value;    // t & 1
      double4_t (&acc)[4] = accs[PAR];
      double4_t (&accn)[4] = accs[1 - PAR];            // result of tile t-1 first, then the start values of tile t+1
      const double* vb = vbuf[t & 1];
      double vs[RPW / 2];                              // the strip is staged in two halves through the same registers
      double2 r[4];                                    // ... and so is the result of tile t-1
      // side operations in issue order: k-tile 0 issues only the loads of the strip (the result of tile t-1 is still
      // leaving the matrix pipe), the k-tiles [1, NKT - 1) share the rest evenly, the last k-tile only prefetches
      constexpr int NE = FIRST ? 0 : 4, NV = STAGE ? 1 : 0;
      // (the loads of tile t+2 go out right behind the image write of tile t+1 -- their registers are free from there --
      //  not at the end of the tile: worth ~1 % of the pass, profiles/r04_pass_layout.txt)
      constexpr int OVA = 0, OW0 = OVA + NV, OE1A = OW0 + NE, OE2A = OE1A + NE, OE1B = OE2A + NE, OE2B = OE1B + NE,
                    OC = OE2B + NE, OVB = OC, ON1 = OVB + NV, ON3 = ON1 + 8, ON2 = ON3 + 8, OWB = ON2 + 16, NSIDE = OWB + NV;
      const __amdgpu_buffer_rsrc_t rsP = rs_rsrc(Pb);
      const unsigned off_prev = tile_off(t - 1);       // tile t-1 (FIRST: unused)
#ifdef RS_SKIP_PMEM
      const bool ok2 = false;
#else
      const bool ok2 = t + 2 < Sw;
#endif
      const __amdgpu_buffer_rsrc_t rs2 = rs_rsrc(ok2 ? (const void*)Pb : (const void*)so);
      const unsigned off2 = ok2 ? tile_off(t + 2) : 0u, ld8d = ok2 ? pl8 : 0u, loffd = ok2 ? loff : lane8 * 2u;
      double* vnext = vbuf[(t + 1) & 1];
      auto side = [&](int o) {
        if constexpr (STAGE) {
          if (o == OVA) stage_load(t + 1, I0_{}, IH_{}, vs);
          if (o == OVB) {
            stage_store(vnext, I0_{}, IH_{}, vs);
            stage_load(t + 1, IH_{}, IH_{}, vs);
          }
          if (o == OWB) stage_store(vnext, IH_{}, IH_{}, vs);
        }
        if constexpr (!FIRST) {                        // result of tile t-1: accumulators -> image -> row-major -> HBM
          if (o >= OW0 && o < OE1A) {
            const int ct = o - OW0;
            if (ct == 0) RS_CBAR();
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) T[cd_index(ct, rg)] = accn[ct][rg];
            if (ct == 3) RS_CBAR();
          }
          if (o >= OE1A && o < OE2A) r[o - OE1A] = *reinterpret_cast<const double2*>(&T[rm_base + 128 * (o - OE1A)]);
          if (o >= OE2A && o < OE1B) stb16<NT>(rsP, loff, off_prev + (unsigned)(o - OE2A) * 2u * pl8, r[o - OE2A]);
          if (o >= OE1B && o < OE2B) r[o - OE1B] = *reinterpret_cast<const double2*>(&T[rm_base + 128 * (o - OE1B + 4)]);
          if (o >= OE2B && o < OC) stb16<NT>(rsP, loff, off_prev + (unsigned)(o - OE2B + 4) * 2u * pl8, r[o - OE2B]);
        }
        if (o >= ON1 && o < ON1 + 8) {                 // tile t+1: row-major registers -> image
          const int q = o - ON1;
          if (q == 0) RS_CBAR();
          *reinterpret_cast<double2*>(&T[rm_base + 128 * q]) = g[q];
          if (q == 7) RS_CBAR();
        } else if (o >= ON2 && o < ON2 + 16) {         // ... -> C/D layout
          const int e = o - ON2;
          accn[e >> 2][e & 3] = T[cd_index(e >> 2, e & 3)];
          if (e == 15) RS_CBAR();
        } else if (o >= ON3 && o < ON3 + 8) {          // tile t+2: HBM -> row-major registers
          const int q = o - ON3;
          g[q] = ldb16<NT>(rs2, loffd, off2 + (unsigned)q * 2u * ld8d);
        }
      };
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        if (kt + 1 < NKT) {
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) bf[(kt + 1) & 1][ct] = vb[((kt + 1) * 4 + ct) * 64 + lane];
        } else if (STAGE) {                            // first fragments of the next tile (published by the barrier below)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) bf[0][ct] = vnext[ct * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);             // (the scheduler would sink the reads below the MFMAs to reuse registers)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#ifdef RS_SKIP_MFMA                                     /* diagnostic build: the memory side alone */
          acc[ct][0] += wf[kt] * bf[kt & 1][ct];
#else
          acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[kt], bf[kt & 1][ct], acc[ct], 0, 0, 0);
#endif
        if (kt == 0) {
          if (NV) side(OVA);
        } else if (kt < NKT - 1) {
#pragma unroll
          for (int o = NV + (kt - 1) * (NSIDE - NV) / (NKT - 2); o < NV + kt * (NSIDE - NV) / (NKT - 2); ++o) side(o);
        }
        if (STAGE && kt == NKT - 2) wg_barrier();      // strip t+1 complete; every B fragment of strip t has been read
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // ---- drain: the last result of this wave (still in its accumulators) ----
    auto drain = [&](auto par_tag) {
      constexpr int PAR = decltype(par_tag)::value;
      double4_t (&acc)[4] = accs[PAR];
      if (i0w == 0 && has_diag) {                      // (uniform) the tile at (0, 0): pose-block noise accumulated since the last pass
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int row = lq + 4 * rg;
          if (row < 3 && li == row) acc[0][rg] += (row == 0) ? dd0 : ((row == 1) ? dd1 : dd2);
        }
      }
      RS_CBAR();
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) T[cd_index(ct, rg)] = acc[ct][rg];
      RS_CBAR();
      const __amdgpu_buffer_rsrc_t rsP = rs_rsrc(Pb);
      const unsigned off_last = tile_off(Sw - 1);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const double2 o = *reinterpret_cast<const double2*>(&T[rm_base + 128 * q]);
        stb16<NT>(rsP, loff, off_last + (unsigned)q * 2u * pl8, o);
      }
      RS_CBAR();
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    if (S == 1) {
      body(T_{}, F_{}, P0{}, 0);
    } else {
      body(T_{}, T_{}, P0{}, 0);
      RS_STAMP(7);
      int t = 1;
      for (; t + 1 < S - 1; t += 2) {
        body(F_{}, T_{}, P1{}, t);
        if (t == 1) RS_STAMP(8);
        body(F_{}, T_{}, P0{}, t + 1);
        if (t == 1) RS_STAMP(9);
        if (t == 3) RS_STAMP(10);
        if (t == 5) RS_STAMP(11);
      }
      RS_STAMP(12);
      if (t < S - 1) {                                 // (t is odd here)
        body(F_{}, T_{}, P1{}, t);
        ++t;
      }
      if (Sw == S) {                                   // this wave also has the unit's last strip: no staging after it
        if ((S - 1) & 1) body(F_{}, F_{}, P1{}, S - 1);
        else body(F_{}, F_{}, P0{}, S - 1);
      }
    }
    RS_STAMP(13);
    if ((Sw - 1) & 1) drain(P1{});
    else drain(P0{});
    RS_STAMP(14);
#ifdef RS_STAMPS
    if (wave == 0 && lane == 0 && unit_no < 3) stamp_buf[unit_no * 20 + 15] = (unsigned long long)S;
#endif
  }
}

// ---------------------------------------------------------------------------------------------
// k_predict_rc: prediction with no observation and nothing pending touches only rows/cols 0,1 and
// the pose diagonal (src/replay_no_ros.py:428-430 with G_F = I outside the 3x3 block).  O(n).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_predict_rc(double* __restrict__ P,
                                                    const double* __restrict__ mu_in,
                                                    double* __restrict__ mu_out,
                                                    const int* __restrict__ nact,
                                                    const SolveOut* __restrict__ so, int ld,
                                                    long pstride) {
  const int b = blockIdx.y;
  const int n = nact[b];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  double* Pb = P + (long)b * pstride;
  const double g0 = so[b].g[0], g1 = so[b].g[1];
  if (j >= 3) {
    mu_out[(long)b * ld + j] = mu_in[(long)b * ld + j];     // k_solve wrote the pose entries
    if (j >= so[b].neff) return;                            // P(2, j) = 0 beyond the active bound
    double* pj = Pb + p_col(ld, j);                         // entry (0, j)
    const int lds = p_lds(ld);
    const double r2 = pj[2 * (long)lds];                    // rows 0,1 of the upper triangle carry the mirrored
    pj[0] += g0 * r2;                                       // column op too
    pj[lds] += g1 * r2;
  } else if (j == 0) {
    double X[3][3], Y[3][3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) X[r][c] = Pb[p_index(ld, min(r, c), max(r, c))];
    for (int c = 0; c < 3; ++c) {
      X[0][c] += g0 * X[2][c];
      X[1][c] += g1 * X[2][c];
    }
    for (int r = 0; r < 3; ++r) {
      Y[r][0] = X[r][0] + g0 * X[r][2];
      Y[r][1] = X[r][1] + g1 * X[r][2];
      Y[r][2] = X[r][2];
      Y[r][r] += so[b].rd[r];
    }
    for (int r = 0; r < 3; ++r)
      for (int c = r; c < 3; ++c) Pb[p_index(ld, r, c)] = Y[r][c];
  }
}

// ---------------------------------------------------------------------------------------------
// k_mirror: lower triangle <- transpose of the upper one (before a download or a dense product; the
// step kernels never read below the diagonal).  64 x 64 tiles through LDS, both sides coalesced.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mirror(double* __restrict__ P, const int* __restrict__ nact, int ld,
                                                long pstride) {
  __shared__ double T[64][65];
  const int b = blockIdx.z;
  const int n = nact[b];
  const int ti = blockIdx.y, tj = blockIdx.x;           // destination tile (rows ti, cols tj), tj <= ti
  if (tj > ti || ti * 64 >= n) return;
  double* Pb = P + (long)b * pstride;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {                    // source tile (rows tj, cols ti)
    const int row = tj * 64 + r, col = ti * 64 + tx;
    T[r][tx] = (row < n && col < n) ? Pb[p_index(ld, row, col)] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int row = ti * 64 + r, col = tj * 64 + tx;
    if (row < n && col < n && col < row) Pb[p_index(ld, row, col)] = T[tx][r];
  }
}

// ---------------------------------------------------------------------------------------------
// k_associate: association, gate, averaging and augmentation of one window on the device
// (src/replay_no_ros.py:280-360), one workgroup (one wave) per trajectory.  The reference's dictionaries have no limits
// (:280-301) and its normal mode hands over a 0.7 s window of every camera frame (:17: 21 frames x every tag in view), so the
// detections are walked in chunks of 64 -- up to EKF_DMAX = 256 -- and up to EKF_AMAX = 32 distinct tags are taken (two
// update passes of EKF_MMAX landmarks: step_out holds two records per trajectory, [0][b] with the prediction and the first
// 16 landmarks, [1][b] the rest).  Per chunk every lane fetches ITS detection (id, the id's landmark index from the tag
// table, IGNORE_TAGS, the gate) -- one memory round trip for 64 detections --; what depends on the ORDER of the detections
// (landmark indices in order of first appearance, :294-295; the update order, :436) is then resolved detection by
// detection without touching memory: the slot of a tag already seen in this window is found by all lanes at once.  The
// per-tag arithmetic and the zeroing of new rows / columns are spread over the lanes.  Writes the StepIn records the solve
// kernel consumes, the tags_positions record, the new state size and the active bound.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_associate(const DetIn* __restrict__ det, int* __restrict__ tagmap,
                                                  int* __restrict__ nact, int* __restrict__ neff_dev,
                                                  double* __restrict__ mu, double* __restrict__ P,
                                                  double* __restrict__ V, double* __restrict__ W,
                                                  StepIn* __restrict__ step_out,
                                                  AssocOut* __restrict__ assoc_out,
                                                  unsigned* __restrict__ flags, AssocConfig cfg, int ld,
                                                  long pstride, int n_max, int pending_k, int batch) {
#pragma clang fp contract(off)
  const int b = blockIdx.x, lane = threadIdx.x;
  const DetIn& d = det[b];
  int* tm = tagmap + (long)b * TAGMAX;
  double* mub = mu + (long)b * ld;
  double* Pb = P + (long)b * pstride;
  double* Vb = V + (long)b * KTOT * ld;
  double* Wb = W + (long)b * KTOT * ld;
  __shared__ int s_idx[AMAX], s_tag[AMAX], s_cnt[AMAX];
  __shared__ double s_t[AMAX][3], s_err[AMAX];
  __shared__ int c_id[64], c_lm[64], c_ok[64];         // the chunk: tag id, its landmark index so far (-1: none), 1 = taken / 2 = bad id
  __shared__ double c_t[64][3], c_e[64];
  const int n_old = nact[b];
  const int nl_old = (n_old - 3) / 2;
  int nl = nl_old, m = 0;                              // (uniform: every lane walks the same sequence)
  unsigned bad = 0;
  const int count = min(d.count, DMAX);
  if (lane < AMAX) {
    s_idx[lane] = -1;
    s_tag[lane] = -1;
    s_cnt[lane] = 0;
    s_t[lane][0] = s_t[lane][1] = s_t[lane][2] = 0.0;
    s_err[lane] = 0.0;
  }
  for (int q0 = 0; q0 < count; q0 += 64) {
    const int q = q0 + lane;
    int id = -1, lm = -1, ok = 0;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, er = 0.0;
    if (q < count) {
      id = d.tag_id[q];
      t0 = d.pose_t[q][0];
      t1 = d.pose_t[q][1];
      t2 = d.pose_t[q][2];
      er = d.pose_err[q];
      if (id < 0 || id >= TAGMAX) ok = 2;
      else {
        bool ign = false;
        for (int u = 0; u < cfg.n_ignore; ++u) ign |= (cfg.ignore[u] == id);       // :286
        if (!ign && !(t2 * t2 + t0 * t0 > cfg.gate2)) {                            // :289
          ok = 1;
          lm = tm[id];
        }
      }
    }
    WAVE_LDS_SYNC();                                   // (the previous chunk's walk has read its entries)
    c_id[lane] = id;
    c_lm[lane] = lm;
    c_ok[lane] = ok;
    c_t[lane][0] = t0;
    c_t[lane][1] = t1;
    c_t[lane][2] = t2;
    c_e[lane] = er;
    WAVE_LDS_SYNC();
    const int nq = min(64, count - q0);
    for (int u = 0; u < nq; ++u) {                     // in order: what a detection gets depends on the ones before it
      const int ok_u = c_ok[u];
      if (ok_u == 2) { bad |= EKF_FLAG_ASSOC; continue; }
      if (ok_u == 0) continue;
      const int id_u = c_id[u];
      const unsigned long long hit = __ballot(lane < m && s_tag[lane] == id_u);    // the tag's slot in this window, if it has one
      int slot = hit ? __builtin_ctzll(hit) : -1;
      if (slot < 0) {
        // a detection that cannot be taken is dropped BEFORE it gets a landmark index: an index handed out here
        // with no measurement behind it would leave an uninitialised landmark in the state
        if (m >= AMAX) { bad |= EKF_FLAG_ASSOC; continue; }
        int lm_u = c_lm[u];                            // (from the table as it stood before this window: a tag new in this
        if (lm_u < 0) {                                //  window has a slot from its first detection on and never comes here again)
          if (3 + 2 * (nl + 1) > n_max) { bad |= EKF_FLAG_ASSOC; continue; }        // :294-295
          lm_u = nl++;
          if (lane == 0) tm[id_u] = lm_u;
        }
        slot = m++;
        if (lane == 0) {
          s_idx[slot] = lm_u;
          s_tag[slot] = id_u;
        }
      }
      if (lane == 0) {
        s_t[slot][0] += c_t[u][0];                                                  // np.mean(..., axis=0): :315
        s_t[slot][1] += c_t[u][1];
        s_t[slot][2] += c_t[u][2];
        s_err[slot] += c_e[u];                                                      // :317
        s_cnt[slot] += 1;
      }
      WAVE_LDS_SYNC();                                 // (the next detection's search reads the slots)
    }
  }
  if (bad && lane == 0) atomicOr(&flags[b], bad);
  WAVE_LDS_SYNC();
  const int n_new = 3 + 2 * nl;
  AssocOut& ao = assoc_out[b];
  const double px = mub[0], py = mub[1], pth = mub[2];         // pose BEFORE the prediction (:331-332)
  if (lane < AMAX) {
    StepIn& so = step_out[(long)(lane / MMAX) * batch + b];
    const int sl = lane % MMAX;
    if (lane < m) {
      const double k = (double)s_cnt[lane];
      const double t0 = s_t[lane][0] / k, t2 = s_t[lane][2] / k;
      const double xr = t2, yr = -t0;                                             // :321
      const double rng = sqrt(xr * xr + yr * yr);                                 // :329
      const double brg = atan2(yr, xr);                                           // :330
      const double xw = px + rng * cos(brg + pth), yw = py + rng * sin(brg + pth);
      const int lm = s_idx[lane];
      so.idx[sl] = lm;
      so.range[sl] = rng;
      so.bearing[sl] = brg;
      ao.idx[lane] = lm;
      ao.tag_id[lane] = s_tag[lane];
      ao.xw[lane] = xw;
      ao.yw[lane] = yw;
      ao.err[lane] = s_err[lane] / k;
      ao.range[lane] = rng;
      ao.bearing[lane] = brg;
      if (lm >= nl_old) {                                                         // :359-360
        mub[3 + 2 * lm] = xw;
        mub[4 + 2 * lm] = yw;
      }
    } else {
      so.idx[sl] = 0;
      so.range[sl] = 0.0;
      so.bearing[sl] = 0.0;
    }
  }
  // augmentation (:341-360): zero the new rows/columns, set the new diagonal; pending ranks see zeros
  const int dn = n_new - n_old;
  const int ld16 = ld >> 4;
  for (long e = lane; e < (long)n_new * dn; e += 64) {
    const int r = (int)(e / dn), q = n_old + (int)(e - (long)r * dn);
    Pb[p_index(ld, r, q)] = (r == q) ? cfg.init_var : 0.0;
    if (r < n_old) Pb[p_index(ld, q, r)] = 0.0;
  }
  for (int e = lane; e < pending_k * dn; e += 64) {
    const int k = e / dn, q = n_old + (e - k * dn);
    Vb[(long)k * ld + q] = 0.0;
    Wb[wm_index(ld16, k, q)] = 0.0;
  }
  if (lane == 0) {
    int bound = neff_dev[b];
    for (int u = 0; u < m; ++u) bound = max(bound, 3 + 2 * (s_idx[u] + 1));
    bound = min(bound, n_new);
    neff_dev[b] = bound;
    nact[b] = n_new;
    StepIn& s0 = step_out[b];
    StepIn& s1 = step_out[(long)batch + b];
    s0.lin = d.lin;
    s0.ang = d.ang;
    s0.m = min(m, MMAX);
    s0.flags = FLAG_PREDICT | FLAG_UPDATE;
    s0.neff = cfg.active_bound ? bound : n_new;
    s0.pad = 0;
    s1.lin = 0.0;                                      // the landmarks beyond the first pass: an update without a prediction
    s1.ang = 0.0;
    s1.m = max(m - MMAX, 0);
    s1.flags = FLAG_UPDATE;
    s1.neff = s0.neff;
    s1.pad = 0;
    ao.m = m;
    ao.n_after = n_new;
  }
}

void launch_associate(hipStream_t st, const DetIn* det, int* tagmap, int* nact, int* neff_dev, double* mu, double* P,
                      double* V, double* W, StepIn* step_out, AssocOut* assoc_out, unsigned* flags,
                      const AssocConfig& cfg, int ld, long pstride, int n_max, int pending_k, int batch) {
  hipLaunchKernelGGL(k_associate, dim3(batch), dim3(64), 0, st, det, tagmap, nact, neff_dev, mu, P, V, W, step_out,
                     assoc_out, flags, cfg, ld, pstride, n_max, pending_k, batch);
}

// Augmentation (src/replay_no_ros.py:341-360): zero rows/cols [n_old, n_new), set the new diagonal.
__global__ __launch_bounds__(256) void k_add_landmarks(double* __restrict__ Pb, double* __restrict__ mub,
                                                       int ld, int n_old, int n_new, double var,
                                                       const double* __restrict__ xy) {
  const int k2 = n_new - n_old;
  const long total = (long)n_new * k2;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int r = (int)(e / k2), q = n_old + (int)(e - (long)r * k2);
    Pb[p_index(ld, r, q)] = (r == q) ? var : 0.0;  // new column block (incl. the new corner)
    if (r < n_old) Pb[p_index(ld, q, r)] = 0.0;    // new row block
    if (r == 0) mub[q] = xy[q - n_old];
  }
}

__global__ __launch_bounds__(256) void k_fill_diag(double* __restrict__ Pb, int ld, int n,
                                                   const double* __restrict__ diag) {
  const long total = (long)n * n;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int r = (int)(e / n), c = (int)(e - (long)r * n);
    Pb[p_index(ld, r, c)] = (r == c) ? diag[r] : 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// launchers (called from ekf_api.hip)
// ---------------------------------------------------------------------------------------------
void launch_solve(hipStream_t st, const double* P, const double* V, const double* W, const double* dacc_in,
                  double* dacc_out, const double* mu_in, double* mu_out, const int* nact, const StepIn* in,
                  SolveOut* out, unsigned* flags, double* fac, const int* neff_floor, unsigned* queue,
                  const DeviceConfig& cfg, int ld, long pstride, int batch, int kbase) {
  hipLaunchKernelGGL(k_solve, dim3(batch), dim3(256), 0, st, P, V, W, dacc_in, dacc_out, mu_in, mu_out, nact, in,
                     out, flags, fac, neff_floor, queue, cfg, ld, pstride, kbase);
}

template <int MCAP>
static void launch_panels_t(hipStream_t st, double* P, double* V, double* W, const double* mu_in,
                            double* mu_out, const int* nact, const SolveOut* so, const double* fac, int ld,
                            long pstride, int batch, int n_hi) {
  // throughput form: a workgroup is four independent waves of 64 state indices sharing one staging
  const long waves = (long)((n_hi + 63) / 64) * batch;
  if (waves <= 512)                                     // latency-bound: four waves split the pending ranks of 64 indices
    hipLaunchKernelGGL((k_panels<MCAP, 4, true>), dim3((n_hi + 63) / 64, batch), dim3(256), 0, st, P, V, W, mu_in,
                       mu_out, nact, so, fac, ld, pstride);
  else
    hipLaunchKernelGGL((k_panels<MCAP, 4, false>), dim3((n_hi + 255) / 256, batch), dim3(256), 0, st, P, V, W,
                       mu_in, mu_out, nact, so, fac, ld, pstride);
}
void launch_panels(hipStream_t st, int mcap, double* P, double* V, double* W, const double* mu_in,
                   double* mu_out, const int* nact, const SolveOut* so, const double* fac, int ld, long pstride,
                   int batch, int n_hi) {
  switch (mcap) {
    case 1: launch_panels_t<1>(st, P, V, W, mu_in, mu_out, nact, so, fac, ld, pstride, batch, n_hi); break;
    case 2: launch_panels_t<2>(st, P, V, W, mu_in, mu_out, nact, so, fac, ld, pstride, batch, n_hi); break;
    case 4: launch_panels_t<4>(st, P, V, W, mu_in, mu_out, nact, so, fac, ld, pstride, batch, n_hi); break;
    case 8: launch_panels_t<8>(st, P, V, W, mu_in, mu_out, nact, so, fac, ld, pstride, batch, n_hi); break;
    default: launch_panels_t<16>(st, P, V, W, mu_in, mu_out, nact, so, fac, ld, pstride, batch, n_hi); break;
  }
}

// The throughput shape of the single-launch step: k_panels<.., SPLIT> (see there).
void launch_step_split_tp(hipStream_t st, int mcap, double* P, double* V, double* W, const double* dacc_in,
                          double* dacc_out, const double* mu_in, double* mu_out, const int* nact, const StepIn* in,
                          SolveOut* out, unsigned* flags, double* fac, const int* neff_floor, unsigned* queue,
                          SolveOut* mbox, unsigned* ready, unsigned seq, int publish, const DeviceConfig& cfg, int ld,
                          long pstride, int batch, int n_hi, int kbase) {
  const dim3 grid(1 + (n_hi + 255) / 256, batch);
  SplitArgs sa{dacc_in, dacc_out, in, out, flags, fac, neff_floor, queue, mbox, ready, seq, publish, kbase, cfg};
#define EKF_SPLIT_TP(M)                                                                                              \
  hipLaunchKernelGGL((k_panels_split<M>), grid, dim3(256), 0, st, P, V, W, mu_in, mu_out, nact, ld, pstride, sa)
  switch (mcap) {
    case 1: EKF_SPLIT_TP(1); break;
    case 2: EKF_SPLIT_TP(2); break;
    case 4: EKF_SPLIT_TP(4); break;
    case 8: EKF_SPLIT_TP(8); break;
    default: break;                                    // (16 landmarks per pass: registers for one wave per SIMD only; the caller
  }                                                    //  takes the two-launch path)
#undef EKF_SPLIT_TP
}

// Whether a step of this shape is run as one launch (the latency regime, see k_step_split): while every panel
// workgroup has a CU to itself.  N=2000: 1 trajectory 33.7 k steps/s against 28.0 k with two launches, 4 trajectories
// 79.2 k against 73.6 k, but 8 trajectories (504 panel workgroups) 85.8 k against 98.8 k.
bool step_is_split(int batch, int n_hi, int cus) { return (long)((n_hi + 63) / 64) * batch <= (long)cus; }
void launch_step_split(hipStream_t st, int mcap, double* P, double* V, double* W, const double* dacc_in, double* dacc_out,
                       const double* mu_in, double* mu_out, const int* nact, const StepIn* in, SolveOut* out,
                       unsigned* flags, double* fac, const int* neff_floor, unsigned* queue, unsigned* ready, unsigned seq,
                       int publish, const DeviceConfig& cfg, int ld, long pstride, int batch, int n_hi, int kbase) {
  const dim3 grid(1 + (n_hi + 63) / 64, batch);
#define EKF_SPLIT(M)                                                                                                \
  hipLaunchKernelGGL((k_step_split<M>), grid, dim3(256), 0, st, P, V, W, dacc_in, dacc_out, mu_in, mu_out, nact, in, out, \
                     flags, fac, neff_floor, queue, ready, seq, publish, cfg, ld, pstride, kbase)
  switch (mcap) {
    case 1: EKF_SPLIT(1); break;
    case 2: EKF_SPLIT(2); break;
    case 4: EKF_SPLIT(4); break;
    case 8: EKF_SPLIT(8); break;
    default: EKF_SPLIT(16); break;
  }
#undef EKF_SPLIT
}

template <int NKTM, int NKL, bool NT>
static void launch_flush_t(hipStream_t st, double* P, const double* V, const double* W, const double* dacc,
                           const int* nact, const SolveOut* so, int ld, long pstride, int batch, int n_hi,
                           int nkt, int rows_per_block) {
  const int gx = (n_hi + 255) / 256, gy = (n_hi + rows_per_block - 1) / rows_per_block;
  int total = 0;                                       // workgroups that reach the upper triangle
  for (int by = 0; by < gy; ++by) total += std::max(0, gx - (by * rows_per_block) / 256);
  hipLaunchKernelGGL((k_flush<NKTM, NKL, NT>), dim3(total, 1, batch), dim3(256), 0, st, P, V, W, dacc, nact, so, ld,
                     pstride, nkt, rows_per_block, gx);
}

// streaming = the batch's covariances do not fit the Infinity Cache: nontemporal accesses
void launch_flush(hipStream_t st, bool streaming, double* P, const double* V, const double* W,
                  const double* dacc, const int* nact, const SolveOut* so, int ld, long pstride, int batch,
                  int n_hi, int nkt, int rows_per_block) {
#define EKF_FLUSH(N, L)                                                                                   \
  do {                                                                                                    \
    if (streaming) launch_flush_t<N, L, true>(st, P, V, W, dacc, nact, so, ld, pstride, batch, n_hi, nkt, rows_per_block); \
    else launch_flush_t<N, L, false>(st, P, V, W, dacc, nact, so, ld, pstride, batch, n_hi, nkt, rows_per_block);          \
  } while (0)
  if (nkt <= 4) EKF_FLUSH(4, 0);
  else if (nkt <= 8) EKF_FLUSH(8, 0);
  else if (nkt <= 12) EKF_FLUSH(12, 0);
  else if (nkt <= 16) EKF_FLUSH(16, 0);
  else EKF_FLUSH(15, 5);
#undef EKF_FLUSH
}

// the row-slab form of the pass (k_flush_rs): persistent workgroups, `queue` = 8 x RS_QSTRIDE zeroed words
template <int NKT, bool NT, bool PAN>
static void launch_flush_rs_t(hipStream_t st, double* P, const double* V, const double* W, const double* dacc,
                              const int* nact, const SolveOut* so, int ld, long pstride, int batch, int n_hi, int nkt,
                              int workgroups, unsigned* queue, int chunk, const int* shares, const CadOut* wv) {
  const int nrb = (n_hi + RS_ROWS - 1) / RS_ROWS;
  if (shares) {                                        // equal static shares (mode 4): one per workgroup
    if (wv)
      hipLaunchKernelGGL((k_flush_rs<NKT, NT, PAN, true>), dim3((unsigned)workgroups), dim3(512), 0, st, P, V, W, dacc, nact, so, ld,
                         pstride, nkt, batch, nrb, 1, 0, 4, queue, shares, wv);
    else
      hipLaunchKernelGGL((k_flush_rs<NKT, NT, PAN, false>), dim3((unsigned)workgroups), dim3(512), 0, st, P, V, W, dacc, nact, so, ld,
                         pstride, nkt, batch, nrb, 1, 0, 4, queue, shares, wv);
    return;
  }
  // Units (see the three modes at k_flush_rs's `pop`).  With an even number of trajectories per queue whole slabs taken
  // trajectory by trajectory balance perfectly (16 trajectories: 366 us against 430 us in chunks of 22 strips); the
  // last trajectory of an odd number finds no partner and its slabs -- only they -- are cut in two (N=2000: 8
  // trajectories 203 us against 256 us with the former rule and 328 us whole; 24: 557 us against 668 us); a batch that
  // is no multiple of 8 would leave the queues with unequal work, so its last trajectories are dealt over all queues
  // slab by slab -- profiles/r02_chunk_sweep.txt, profiles/r02_batch_sweep.txt.  Below 8 trajectories (the row-slab
  // kernel then only runs for long ones, N=8000) every slab is cut so that there are about three units per CU.
  const int s_max = (n_hi + 63) / 64;                  // strips of the longest slab
  int cs = s_max, nch = 1, mode = 0;
  if (chunk > 0) {                                     // ("pass_chunk" option: every slab of every trajectory)
    cs = std::min(std::max(chunk, 1), s_max);
    nch = (s_max + cs - 1) / cs;
  } else if (batch >= 8) {
    if (batch % 8 == 0) {                              // pairs; an unpaired trajectory (8: each queue's only one) cut in two
      mode = 1;
      if (s_max >= 4) {
        cs = (s_max + 1) / 2;
        nch = 2;
      }
    } else if (batch <= 12 && s_max >= 8) {             // (N=2000: 9 trajectories 253 us against 329 us whole, 10: 284 / 333,
                                                       //  12: 327 / 342; 14: 360 / 344 -- from 13 on whole slabs)
      mode = 3;                                        // equal work per queue, half slabs, longest first
      nch = (s_max + 3) / 4;                           // h: half a chunk; chunks of cs = 2 h strips
      cs = 2 * nch;
    } else {
      mode = 2;                                        // equal work per queue, longest slabs first
    }
  } else {                                             // a few long trajectories (N=8000): about three units per CU
    long steps = 0;
    for (int rb = 0; rb < nrb; ++rb) steps += std::max(1, s_max - 2 * rb);
    steps *= batch;
    if ((long)nrb * batch < 3L * workgroups) {
      cs = (int)std::max<long>(2, (steps + 3L * workgroups - 1) / (3L * workgroups));
      nch = (s_max + cs - 1) / cs;
    }
  }
  const long units = (long)nrb * (mode == 0 ? nch : mode == 3 ? 2 : 1) * batch;   // (modes 1, 2: at least; only the grid size depends on it)
  if (wv)
    hipLaunchKernelGGL((k_flush_rs<NKT, NT, PAN, true>), dim3((unsigned)std::min<long>(workgroups, units)), dim3(512), 0, st, P, V, W,
                       dacc, nact, so, ld, pstride, nkt, batch, nrb, nch, cs, mode, queue, nullptr, wv);
  else
    hipLaunchKernelGGL((k_flush_rs<NKT, NT, PAN, false>), dim3((unsigned)std::min<long>(workgroups, units)), dim3(512), 0, st, P, V, W,
                       dacc, nact, so, ld, pstride, nkt, batch, nrb, nch, cs, mode, queue, nullptr, wv);
}

void launch_flush_rs(hipStream_t st, bool streaming, double* P, const double* V, const double* W, const double* dacc,
                     const int* nact, const SolveOut* so, int ld, long pstride, int batch, int n_hi, int nkt,
                     int workgroups, unsigned* queue, int chunk, const int* shares, const CadOut* wv) {
#define EKF_FLUSH_RS(N)                                                                                   \
  do {                                                                                                    \
    if (ld > PPW) {                                                                                       \
      if (streaming) launch_flush_rs_t<N, true, true>(st, P, V, W, dacc, nact, so, ld, pstride, batch, n_hi, nkt, workgroups, queue, chunk, shares, wv); \
      else launch_flush_rs_t<N, false, true>(st, P, V, W, dacc, nact, so, ld, pstride, batch, n_hi, nkt, workgroups, queue, chunk, shares, wv);          \
    } else {                                                                                              \
      if (streaming) launch_flush_rs_t<N, true, false>(st, P, V, W, dacc, nact, so, ld, pstride, batch, n_hi, nkt, workgroups, queue, chunk, shares, wv); \
      else launch_flush_rs_t<N, false, false>(st, P, V, W, dacc, nact, so, ld, pstride, batch, n_hi, nkt, workgroups, queue, chunk, shares, wv);          \
    }                                                                                                     \
  } while (0)
  if (nkt <= 4) EKF_FLUSH_RS(4);
  else if (nkt <= 8) EKF_FLUSH_RS(8);
  else if (nkt <= 12) EKF_FLUSH_RS(12);
  else if (nkt <= 16) EKF_FLUSH_RS(16);
  else EKF_FLUSH_RS(20);
#undef EKF_FLUSH_RS
}

#ifdef RS_STAMPS
int flush_rs_queue_words() { return 8 * RS_QSTRIDE + 256 * 64 * 2; }
#else
int flush_rs_queue_words() { return 8 * RS_QSTRIDE; }
#endif

void launch_predict_rc(hipStream_t st, double* P, const double* mu_in, double* mu_out, const int* nact,
                       const SolveOut* so, int ld, long pstride, int batch, int n_hi) {
  hipLaunchKernelGGL(k_predict_rc, dim3((n_hi + 255) / 256, batch), dim3(256), 0, st, P, mu_in, mu_out,
                     nact, so, ld, pstride);
}

// ---------------------------------------------------------------------------------------------
// k_pack_small: the download of a SMALL state (n <= PACK_SMALL_N) as one kernel: the mirrored n x n covariance (read
// from the stored upper triangle), the mean and the trajectory's sticky flags written straight into pinned host memory
// (zero copy) -- one launch and one synchronisation instead of a mirror launch, two device-to-host copies (~10 us each
// for a few KB) and a separate copy of the flags.  What the reference's loop pays per step at its real map size
// (12 landmarks: src/replay_no_ros.py:26, :229-237) is this latency, not bandwidth.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_small(const double* __restrict__ Pb, const double* __restrict__ mub,
                                                    const unsigned* __restrict__ flag_b, int ld, int n,
                                                    double* __restrict__ host_out) {
  const int total = n * n;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int r = e / n, c = e - r * n;
    host_out[e] = Pb[p_index(ld, min(r, c), max(r, c))];
  }
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < n; i += 256) host_out[total + i] = mub[i];
    if (threadIdx.x == 0) host_out[total + n] = (double)(*flag_b);
  }
}
void launch_pack_small(hipStream_t st, const double* Pb, const double* mub, const unsigned* flag_b, int ld, int n,
                       double* host_out) {
  const int blocks = std::min((n * n + 255) / 256, 64);
  hipLaunchKernelGGL(k_pack_small, dim3(blocks), dim3(256), 0, st, Pb, mub, flag_b, ld, n, host_out);
}

// ---------------------------------------------------------------------------------------------
// k_pack_dense: the download of a LARGE state into pinned (device-visible) host memory as one kernel: every 64 x 64
// tile of the stored upper triangle is read once and written twice -- as it stands and, through LDS, transposed --
// straight into the caller's dense n x n array, 512-byte row segments both times.  No mirror pass over the device
// copy, and no SDMA copy: the runtime's choice of copy engine halves the rate of an 8 MB rectangle copy once a
// process has used more streams (profiles/r04_small_state.txt, part 3); stores from 2016 workgroups do not care.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_dense(const double* __restrict__ Pb, int ld, int n, double* __restrict__ host_out) {
  __shared__ double T[64][65];
  // tile (ti, tj), ti <= tj, from the linear workgroup index (row ti holds nt - ti tiles)
  const int nt = (n + 63) >> 6;
  int ti = 0, rest = blockIdx.x;
  while (rest >= nt - ti) {
    rest -= nt - ti;
    ++ti;
  }
  const int tj = ti + rest;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int row = ti * 64 + r, col = tj * 64 + tx;
    double v = 0.0;
    if (row < n && col < n) v = Pb[p_index(ld, min(row, col), max(row, col))];     // (diagonal tile: its lower half mirrored)
    T[r][tx] = v;
  }
  __syncthreads();
  // over PCIe in 16-byte stores, two 512-byte row segments per wave instruction: lanes 0 - 31 one row, lanes 32 - 63 the next
  const int half = tx >> 5, c2 = 2 * (tx & 31);
#pragma unroll 2
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1 && ti == tj) break;
    for (int r = 2 * ty + half; r < 64; r += 8) {
      const int row = (pass ? tj : ti) * 64 + r, col = (pass ? ti : tj) * 64 + c2;
      if (row >= n || col >= n) continue;
      const double v0 = pass ? T[c2][r] : T[r][c2], v1 = pass ? T[c2 + 1][r] : T[r][c2 + 1];
      double* dst = host_out + (long)row * n + col;
      if (col + 1 < n) {
        typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
        *reinterpret_cast<d2*>(dst) = d2{v0, v1};
      } else {
        *dst = v0;
      }
    }
  }
}
void launch_pack_dense(hipStream_t st, const double* Pb, int ld, int n, double* host_out) {
  const int nt = (n + 63) / 64;
  hipLaunchKernelGGL(k_pack_dense, dim3(nt * (nt + 1) / 2), dim3(256), 0, st, Pb, ld, n, host_out);
}

void launch_mirror(hipStream_t st, double* P, const int* nact, int ld, long pstride, int batch, int n_hi) {
  const int t = (n_hi + 63) / 64;
  hipLaunchKernelGGL(k_mirror, dim3(t, t, batch), dim3(256), 0, st, P, nact, ld, pstride);
}

void launch_add_landmarks(hipStream_t st, double* Pb, double* mub, int ld, int n_old, int n_new,
                          double var, const double* xy) {
  const long total = (long)n_new * (n_new - n_old);
  const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(k_add_landmarks, dim3(blocks), dim3(256), 0, st, Pb, mub, ld, n_old, n_new, var, xy);
}

void launch_fill_diag(hipStream_t st, double* Pb, int ld, int n, const double* diag) {
  const long total = (long)n * n;
  const int blocks = (int)std::min<long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(k_fill_diag, dim3(blocks), dim3(256), 0, st, Pb, ld, n, diag);
}

}  // namespace ekf
