// Hand-written HIP kernels of the EKF-SLAM step for MI355X (gfx950, wave64).
//
// One reference step (src/replay_no_ros.py:363-482: predict, then one dense (I-KH)P product per
// observed landmark) is executed as
//   k_solve    the sequential part on the compressed c x c system (c = 3+2m), one wave per trajectory
//   k_panels   thread i replays the m rank-2 down-dates on column i of the row panel P'[C,:] and on
//              row i of the column panel P'[:,C]:  V = stacked H_j P_j (2m x n, coalesced reads),
//              W = -stacked K_j (n x 2m); mean update
//   k_pass     P <- P + Rt + W V   one streaming read-modify-write of P (HBM-bound, 16 n^2 bytes)
// or, with no observation, k_predict_rc (rows/cols 0,1 of P only, O(n)).
// The algebra is restated on the CPU in oracle/ekf_oracle.py::ekf_step_structured.
#include <algorithm>

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

namespace ekf {

__device__ __forceinline__ double wrap_pi(double a) {
  // (a + pi) % (2 pi) - pi with NumPy remainder semantics, result in [-pi, pi)
  // (src/replay_no_ros.py:397, :458).  fma(-k, 2pi, x) is the exact remainder when k is the right
  // quotient (the remainder is representable); the two fix-ups cover a quotient that is off by one.
  const double two_pi = 2.0 * M_PI;
  const double x = a + M_PI;
  const double k = floor(x * (1.0 / two_pi));
  double r = fma(-k, two_pi, x);
  if (r < 0.0) r += two_pi;
  else if (r >= two_pi) r -= two_pi;
  return r - M_PI;
}

// Innovation and 2x5 Jacobian of one range/bearing observation (src/replay_no_ros.py:443-469).
// h[r][k] = row r, column k on {x, y, theta, lx, ly}.  q == 0 gives NaN rows like NumPy's 0/0.
__device__ __forceinline__ void linearize(const double* muc, int a, double z_range, double z_bearing,
                                          double (&h)[2][5], double& y0, double& y1) {
  const double dx = muc[a] - muc[0], dy = muc[a + 1] - muc[1];   // :443
  const double th = muc[2];
  const double q = dx * dx + dy * dy;                             // :446
  // 1/sqrt(q) once (hardware estimate + two Newton steps, <= 1 ulp); sqrt(q) = q * rs, 1/q = rs * rs.
  // q == 0 -> rs = inf -> NaN rows below, like NumPy's 0/0 at :466-469.
  double rs = __builtin_amdgcn_rsq(q);
  rs = rs * fma(-0.5 * q * rs, rs, 1.5);
  rs = rs * fma(-0.5 * q * rs, rs, 1.5);
  const double sq = q * rs;
  const double rq = rs * rs;
  y0 = z_range - sq;                                              // :455
  y1 = wrap_pi(z_bearing - (atan2(dy, dx) - th));                 // :453-458
  const double nanv = __builtin_nan("");
  h[0][0] = -rs * dx;                                             // (-sqrt(q) dx) / q
  h[0][1] = -rs * dy;
  h[0][2] = (q > 0.0) ? 0.0 : nanv;                               // .0 / q
  h[0][3] = rs * dx;
  h[0][4] = rs * dy;
  h[1][0] = dy * rq;
  h[1][1] = -dx * rq;
  h[1][2] = (q > 0.0 && q < __builtin_inf()) ? -1.0 : nanv;       // -q / q
  h[1][3] = -dy * rq;
  h[1][4] = dx * rq;
}

// ---------------------------------------------------------------------------------------------
// k_solve: one 64-lane wave per trajectory; lane l < c owns compressed index l and keeps column l
// of P[C,C] in registers (written through to LDS for the row/column reads of the other lanes).
// The kernel is latency-bound (one wave, m sequential iterations): every phase is written as
// batches of independent operations, and the next landmark's linearisation (atan2, sqrt, 1/q:
// ~0.2 us) is issued beside the covariance down-date, which does not depend on it.
// Reads mu_in, writes mu_out[C] (the mean is double-buffered so that no kernel of a step reads
// an entry another workgroup of the same step writes).
// ---------------------------------------------------------------------------------------------
constexpr int PCS = CMAX + 2;   // LDS row stride 37 doubles: column reads by 32 lanes are conflict-free
constexpr int RCH = 8;          // rows per batch of the down-date
constexpr int CPAD = (CMAX + RCH - 1) / RCH * RCH;   // 40

__global__ __launch_bounds__(64) void k_solve(const double* __restrict__ P,
                                              const double* __restrict__ mu_in,
                                              double* __restrict__ mu_out,
                                              const int* __restrict__ nact,
                                              const StepIn* __restrict__ in,
                                              SolveOut* __restrict__ out,
                                              unsigned* __restrict__ flags, DeviceConfig cfg, int ld,
                                              long pstride) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const StepIn& s = in[b];
  SolveOut& o = out[b];
  const double* Pb = P + (long)b * pstride;

  __shared__ double Pc[CPAD][PCS];
  __shared__ double muc[CPAD];
  __shared__ double2 hpS[CPAD], kcS[CPAD];
  __shared__ int Cs[CPAD];

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
  if (lane < CPAD) {
    Cs[lane] = Cl;
    kcS[lane] = make_double2(0.0, 0.0);
    hpS[lane] = make_double2(0.0, 0.0);
    if (lane < CMAX + 3) o.C[lane] = Cl;
  }
  const double mu_l = mu_in[(long)b * ld + Cl];
  __syncthreads();
  STAMP(o, 1);
  double pcol[CPAD];
  {
    const double* colp = Pb + Cl;
#pragma unroll
    for (int r = 0; r < CPAD; ++r) pcol[r] = (r < 3 || (on && r < c)) ? colp[(long)Cs[r] * ld] : 0.0;
    if (!on) {
#pragma unroll
      for (int r = 0; r < 3; ++r) pcol[r] = 0.0;
    }
  }
  STAMP(o, 2);
  // ---- motion model (src/replay_no_ros.py:368-417), evaluated redundantly by every lane ----
  const double th = __shfl(mu_l, 2);
  double g0 = 0.0, g1 = 0.0, nx = __shfl(mu_l, 0), ny = __shfl(mu_l, 1), nth = th;
  if (do_pred && !cfg.disable_motion_model) {
    const double lin = s.lin, ang = s.ang;
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
  const double rd0 = do_pred ? cfg.rd[0] : 0.0, rd1 = do_pred ? cfg.rd[1] : 0.0,
               rd2 = do_pred ? cfg.rd[2] : 0.0;
  STAMP(o, 3);
  // P'[C,C] = Gc P[C,C] Gc^T + Rt  (:428-430 restricted to C), column `lane` in registers:
  // row ops on rows 0,1; column ops need column 2 of the row-updated matrix (lane 2's registers).
  const double p22 = __shfl(pcol[2], 2);
  pcol[0] += g0 * pcol[2];
  pcol[1] += g1 * pcol[2];
  if (lane == 2) {                                   // X[:,2] after the row ops, for the column ops
#pragma unroll
    for (int r = 0; r < CPAD; ++r) muc[r] = pcol[r];
  }
  __syncthreads();
  if (lane < 2) {
    const double gl = (lane == 0) ? g0 : g1;
#pragma unroll
    for (int r = 0; r < CPAD; ++r) pcol[r] = fma(gl, muc[r], pcol[r]);
  }
  if (lane == 0) pcol[0] += rd0;
  if (lane == 1) pcol[1] += rd1;
  if (lane == 2) pcol[2] += rd2;
  __syncthreads();
  if (on) {
#pragma unroll
    for (int r = 0; r < CPAD; ++r) Pc[r][lane] = pcol[r];
  }
  double mu_cur = (lane == 0) ? nx : ((lane == 1) ? ny : ((lane == 2) ? nth : mu_l));
  if (lane < CPAD) muc[lane] = on ? mu_cur : 0.0;
  if (lane == 0) {
    o.g[0] = g0;
    o.g[1] = g1;
    o.rd[0] = rd0;
    o.rd[1] = rd1;
    o.rd[2] = rd2;
    o.p22h = 0.5 * p22;
    o.c = c;
    o.m = m;
  }
  __syncthreads();

  STAMP(o, 4);
  // ---- sequential per-landmark recurrences (:436-480) on the compressed system ----
  double h[2][5], y0 = 0.0, y1 = 0.0;
  if (m > 0) linearize(muc, 3, s.range[0], s.bearing[0], h, y0, y1);
  STAMP(o, 5);
  for (int j = 0; j < m; ++j) {
    const int a = 3 + 2 * j;
    SolveIter& it = o.it[j];
    STAMP(o, 8 + 6 * j);
    // phase A: rows sel of P_j at column C[lane] (H P) and columns sel at row C[lane] (P H^T).
    // A lone wave issues one fp64 VALU op per ~8 cycles, so the instruction count is what matters:
    // S is formed from the five hp pairs the other lanes publish in LDS (20 FMAs), not recomputed.
    double pr[5], pq[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int sk = (k < 3) ? k : a + (k - 3);
      pr[k] = Pc[sk][ll];
      pq[k] = Pc[ll][sk];
    }
    double hp0 = h[0][0] * pr[0], hp1 = h[1][0] * pr[0], ph0 = pq[0] * h[0][0], ph1 = pq[0] * h[1][0];
#pragma unroll
    for (int k = 1; k < 5; ++k) {
      hp0 = fma(h[0][k], pr[k], hp0);
      hp1 = fma(h[1][k], pr[k], hp1);
      ph0 = fma(pq[k], h[0][k], ph0);
      ph1 = fma(pq[k], h[1][k], ph1);
    }
    if (on) hpS[lane] = make_double2(hp0, hp1);
    if (lane < CMAX) *reinterpret_cast<double2*>(it.hpt[lane]) = on ? make_double2(hp0, hp1) : make_double2(0.0, 0.0);
    __syncthreads();
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
    const double k0 = ph0 * i00 + ph1 * i10;                            // K_j[C[lane], :]
    const double k1 = ph0 * i01 + ph1 * i11;
    if (on) {
      kcS[lane] = make_double2(k0, k1);
      mu_cur += k0 * y0 + k1 * y1;                                      // :476
      muc[lane] = mu_cur;
    }
    if (lane < CMAX) *reinterpret_cast<double2*>(it.kc[lane]) = on ? make_double2(k0, k1) : make_double2(0.0, 0.0);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 5; ++k) *reinterpret_cast<double2*>(it.h5t[k]) = make_double2(h[0][k], h[1][k]);
      *reinterpret_cast<double2*>(&it.si[0]) = make_double2(i00, i01);
      *reinterpret_cast<double2*>(&it.si[2]) = make_double2(i10, i11);
      *reinterpret_cast<double2*>(it.y) = make_double2(y0, y1);
    }
    __syncthreads();
    STAMP(o, 10 + 6 * j);
    // phase C: the gains of every row are fetched first, the next landmark's linearisation (needs
    // only the mean) runs while they land, then the down-date (:480) as batches of independent FMAs
    if (j + 1 < m) {
      double2 kr[CPAD];
#pragma unroll
      for (int r0 = 0; r0 < CPAD; r0 += RCH)
        if (r0 < c) {
#pragma unroll
          for (int u = 0; u < RCH; ++u) kr[r0 + u] = kcS[r0 + u];
        }
      double hn[2][5], yn0, yn1;
      linearize(muc, a + 2, s.range[j + 1], s.bearing[j + 1], hn, yn0, yn1);
      STAMP(o, 11 + 6 * j);
#pragma unroll
      for (int r0 = 0; r0 < CPAD; r0 += RCH)
        if (r0 < c) {
#pragma unroll
          for (int u = 0; u < RCH; ++u) pcol[r0 + u] = fma(-kr[r0 + u].x, hp0, pcol[r0 + u]);
#pragma unroll
          for (int u = 0; u < RCH; ++u) pcol[r0 + u] = fma(-kr[r0 + u].y, hp1, pcol[r0 + u]);
          if (on) {
#pragma unroll
            for (int u = 0; u < RCH; ++u) Pc[r0 + u][lane] = pcol[r0 + u];
          }
        }
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 5; ++k) h[r][k] = hn[r][k];
      y0 = yn0;
      y1 = yn1;
    }
    __syncthreads();
    STAMP(o, 12 + 6 * j);
  }

  STAMP(o, 6);
  bool bad = false;
  if (on) {
    mu_out[(long)b * ld + Cl] = mu_cur;
    bad = !(fabs(mu_cur) <= 1.79769313486231570815e308);
  }
  if (__any(bad) && lane == 0) atomicOr(&flags[b], EKF_FLAG_NONFINITE);
}

// ---------------------------------------------------------------------------------------------
// k_panels: thread i owns column i of the row panel R = P'[C,:] and row i of the column panel
// L = P'[:,C] (both c values in registers) and replays the m sequential rank-2 down-dates on them
// with the per-iteration uniforms of k_solve staged in LDS (16-byte broadcast reads):
//   V[2j..2j+1][i] = H_j P_j[:, i]        = h5_j . R[sel_j]
//   W[i][2j..2j+1] = -K_j[i, :]           = -(L[sel_j] . h5_j^T) S_j^-1
//   R -= K_j[C,:] (H_j P_j)[:, i],   L -= K_j[i,:] (H_j P_j)[:, C]
// plus the two rank-1 pairs of the motion Jacobian (:430) at k = 2*MCAP, 2*MCAP+1:
//   V[2M] = P[2,:] + p22h*gt, W[:,2M] = gt,   V[2M+1] = gt, W[:,2M+1] = P[:,2] + p22h*gt
// and the mean: mu_out[i] = mu_in[i] + sum_j K_j[i,:] y_j for i not in C.
// ---------------------------------------------------------------------------------------------
template <int MCAP>
__global__ __launch_bounds__(64) void k_panels(const double* __restrict__ P,
                                               const double* __restrict__ mu_in,
                                               double* __restrict__ mu_out,
                                               const int* __restrict__ nact,
                                               const SolveOut* __restrict__ so,
                                               double* __restrict__ V, double* __restrict__ W, int ld,
                                               long pstride) {
  constexpr int CC = 3 + 2 * MCAP, K2 = 2 * MCAP;
  constexpr int VS = 2 * MMAX + 2;
  __shared__ SolveIter its[MCAP];
  const int b = blockIdx.y;
  const int n = nact[b];
  if ((int)blockIdx.x * 64 >= n) return;
  const int tid = threadIdx.x;
  const int j = blockIdx.x * 64 + tid;
  const bool act = j < n;
  const int jj = act ? j : 0;
  const SolveOut& o = so[b];
  const double* Pb = P + (long)b * pstride;
  const int m = min(o.m, MCAP), c = o.c;

  {
    const double2* src = reinterpret_cast<const double2*>(o.it);
    double2* dst = reinterpret_cast<double2*>(its);
    const int count = m * (int)(sizeof(SolveIter) / 16);
    for (int t = tid; t < count; t += 64) dst[t] = src[t];
  }
  const double g0 = o.g[0], g1 = o.g[1];
  const double gj = (jj == 0) ? g0 : ((jj == 1) ? g1 : 0.0);

  double R[CC], L[CC];
#pragma unroll
  for (int a = 0; a < CC; ++a) R[a] = Pb[(long)o.C[a] * ld + jj];
#pragma unroll
  for (int a = 0; a < CC; ++a) L[a] = Pb[(long)jj * ld + o.C[a]];
  const double raw_r2 = R[2], raw_c2 = L[2];
  const double mu_i = mu_in[(long)b * ld + jj];

  // P' = G_F P G_F^T + F^T R F  (src/replay_no_ros.py:430) on the two panels
  R[0] += g0 * R[2];
  R[1] += g1 * R[2];
  if (jj < 2) {
    const double p22 = Pb[2 * (long)ld + 2];
#pragma unroll
    for (int a = 0; a < CC; ++a) {
      double x = Pb[(long)o.C[a] * ld + 2];
      if (a == 0) x += g0 * p22;
      if (a == 1) x += g1 * p22;
      R[a] += gj * x;
      L[a] += gj * Pb[2 * (long)ld + o.C[a]];
    }
  }
  L[0] += g0 * L[2];
  L[1] += g1 * L[2];
#pragma unroll
  for (int a = 0; a < 3; ++a)
    if (a == jj) {
      R[a] += o.rd[a];
      L[a] += o.rd[a];
    }
  __syncthreads();

  double* Vb = V + (long)b * VS * ld;
  double* Wr = W + ((long)b * ld + jj) * VS;
  double dm = 0.0;
#pragma unroll
  for (int it = 0; it < MCAP; ++it) {
    if (it < m) {
      const SolveIter& I = its[it];
      const int a0 = 3 + 2 * it;
      const bool more = it + 1 < m;
      // every LDS broadcast read of the iteration is issued up front (one latency exposure)
      double2 hk[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) hk[k] = *reinterpret_cast<const double2*>(I.h5t[k]);
      const double2 s01 = *reinterpret_cast<const double2*>(&I.si[0]);
      const double2 s23 = *reinterpret_cast<const double2*>(&I.si[2]);
      const double2 yy = *reinterpret_cast<const double2*>(I.y);
      constexpr bool PREFETCH = MCAP <= 8;
      double2 kcv[PREFETCH ? CC : 1], hcv[PREFETCH ? CC : 1];
      if (PREFETCH && more) {
#pragma unroll
        for (int a = 0; a < CC; ++a) {
          kcv[a] = *reinterpret_cast<const double2*>(I.kc[a]);
          hcv[a] = *reinterpret_cast<const double2*>(I.hpt[a]);
        }
      }
      double hp0 = hk[0].x * R[0], hp1 = hk[0].y * R[0], ph0 = L[0] * hk[0].x, ph1 = L[0] * hk[0].y;
#pragma unroll
      for (int k = 1; k < 5; ++k) {
        const double rv = (k < 3) ? R[k] : R[a0 + (k - 3)];
        const double lv = (k < 3) ? L[k] : L[a0 + (k - 3)];
        hp0 = fma(hk[k].x, rv, hp0);
        hp1 = fma(hk[k].y, rv, hp1);
        ph0 = fma(lv, hk[k].x, ph0);
        ph1 = fma(lv, hk[k].y, ph1);
      }
      const double k0 = ph0 * s01.x + ph1 * s23.x;
      const double k1 = ph0 * s01.y + ph1 * s23.y;
      dm += k0 * yy.x + k1 * yy.y;
      if (act) {
        Vb[(long)(2 * it) * ld + j] = hp0;
        Vb[(long)(2 * it + 1) * ld + j] = hp1;
        Wr[2 * it] = -k0;
        Wr[2 * it + 1] = -k1;
      }
      if (more) {
        if (PREFETCH) {
#pragma unroll
          for (int a = 0; a < CC; ++a) {
            R[a] = fma(-kcv[a].x, hp0, R[a]);
            L[a] = fma(-k0, hcv[a].x, L[a]);
          }
#pragma unroll
          for (int a = 0; a < CC; ++a) {
            R[a] = fma(-kcv[a].y, hp1, R[a]);
            L[a] = fma(-k1, hcv[a].y, L[a]);
          }
        } else {
#pragma unroll
          for (int c0 = 0; c0 < CC; c0 += 8) {
            double2 kc[8], hc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (c0 + u < CC) {
                kc[u] = *reinterpret_cast<const double2*>(I.kc[c0 + u]);
                hc[u] = *reinterpret_cast<const double2*>(I.hpt[c0 + u]);
              }
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (c0 + u < CC) {
                R[c0 + u] = fma(-kc[u].x, hp0, R[c0 + u]);
                L[c0 + u] = fma(-k0, hc[u].x, L[c0 + u]);
              }
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (c0 + u < CC) {
                R[c0 + u] = fma(-kc[u].y, hp1, R[c0 + u]);
                L[c0 + u] = fma(-k1, hc[u].y, L[c0 + u]);
              }
          }
        }
      }
    } else if (act) {
      Vb[(long)(2 * it) * ld + j] = 0.0;
      Vb[(long)(2 * it + 1) * ld + j] = 0.0;
      Wr[2 * it] = 0.0;
      Wr[2 * it + 1] = 0.0;
    }
  }
  if (act) {
    Vb[(long)K2 * ld + j] = raw_r2 + o.p22h * gj;
    Vb[(long)(K2 + 1) * ld + j] = gj;
    Wr[K2] = gj;
    Wr[K2 + 1] = raw_c2 + o.p22h * gj;
    bool inC = false;
    for (int a = 0; a < c; ++a) inC |= (o.C[a] == j);
    if (!inC) mu_out[(long)b * ld + j] = mu_i + dm;
  }
}

// ---------------------------------------------------------------------------------------------
// k_pass: P[i][j] += Rt + sum_{k<KT} W[i][k] V[k][j], in place, one read + one write of P.
// A wave owns a strip of 128 columns (2 adjacent doubles per lane = one 1 KiB row segment per
// load/store instruction); its V strip lives in registers for the whole row block, W[i][:] is
// wave-uniform and comes through the scalar cache.  UNR rows are in flight per wave.
// NT: nontemporal loads AND stores -- for working sets beyond the 256 MiB Infinity Cache the pair
// is worth +25 % (5.7-6.0 vs 4.5 TB/s measured); for a resident P plain accesses are faster.
// The odd last column (n = 3+2N is odd) is done by the wave whose strip holds it, one row per lane.
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

template <int MCAP, int UNR, int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * 64) void k_pass(double* __restrict__ P,
                                                     const double* __restrict__ V,
                                                     const double* __restrict__ W,
                                                     const int* __restrict__ nact,
                                                     const SolveOut* __restrict__ so, int ld,
                                                     long pstride, int rows_per_block) {
  constexpr int KT = 2 * MCAP + 2;
  constexpr int WS = 2 * MMAX + 2;
  const int b = blockIdx.z;
  const int n = nact[b];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int strip = blockIdx.x * WAVES + wave;
  const int i0 = blockIdx.y * rows_per_block;
  if (strip * 128 >= n || i0 >= n) return;
  const int i1 = min(n, i0 + rows_per_block);
  const int j0 = strip * 128 + lane * 2;

  double* Pb = P + (long)b * pstride;
  const double* Vb = V + (long)b * WS * ld;
  const double* Wb = W + (long)b * ld * WS;

  if (j0 + 1 < n) {
    double2 v[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) v[k] = *reinterpret_cast<const double2*>(Vb + (long)k * ld + j0);
    int i = i0;
    for (; i + UNR <= i1; i += UNR) {
      double2 p[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) p[u] = ld2<NT>(Pb + (long)(i + u) * ld + j0);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const double* w = Wb + (long)(i + u) * WS;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const double wk = w[k];
          p[u].x += wk * v[k].x;
          p[u].y += wk * v[k].y;
        }
      }
      if (i < 3) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int r = i + u;
          if (r < 3) {
            if (j0 == r) p[u].x += so[b].rd[r];
            if (j0 + 1 == r) p[u].y += so[b].rd[r];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) st2<NT>(Pb + (long)(i + u) * ld + j0, p[u]);
    }
    for (; i < i1; ++i) {
      double2 p = ld2<NT>(Pb + (long)i * ld + j0);
      const double* w = Wb + (long)i * WS;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const double wk = w[k];
        p.x += wk * v[k].x;
        p.y += wk * v[k].y;
      }
      if (i < 3) {
        if (j0 == i) p.x += so[b].rd[i];
        if (j0 + 1 == i) p.y += so[b].rd[i];
      }
      st2<NT>(Pb + (long)i * ld + j0, p);
    }
  }
  // odd last column: one row per lane
  const int tail = n - 1;
  if ((n & 1) && tail >= strip * 128 && tail < strip * 128 + 128) {
    for (int r = i0 + lane; r < i1; r += 64) {
      double p = Pb[(long)r * ld + tail];
      const double* w = Wb + (long)r * WS;
#pragma unroll
      for (int k = 0; k < KT; ++k) p += w[k] * Vb[(long)k * ld + tail];
      if (r < 3 && r == tail) p += so[b].rd[r];
      Pb[(long)r * ld + tail] = p;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_predict_rc: prediction with no observation touches only rows/cols 0,1 and the pose diagonal
// (src/replay_no_ros.py:428-430 with G_F = I outside the 3x3 block).  O(n).
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
    const double r2 = Pb[2 * (long)ld + j];
    Pb[j] += g0 * r2;
    Pb[(long)ld + j] += g1 * r2;
    const double c2 = Pb[(long)j * ld + 2];
    Pb[(long)j * ld + 0] += g0 * c2;
    Pb[(long)j * ld + 1] += g1 * c2;
  } else if (j == 0) {
    double X[3][3], Y[3][3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) X[r][c] = Pb[(long)r * ld + c];
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
      for (int c = 0; c < 3; ++c) Pb[(long)r * ld + c] = Y[r][c];
  }
}

// Augmentation (src/replay_no_ros.py:341-360): zero rows/cols [n_old, n_new), set the new diagonal.
__global__ __launch_bounds__(256) void k_add_landmarks(double* __restrict__ Pb, double* __restrict__ mub,
                                                       int ld, int n_old, int n_new, double var,
                                                       const double* __restrict__ xy) {
  const int k2 = n_new - n_old;
  const long total = (long)n_new * k2;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int r = (int)(e / k2), q = n_old + (int)(e - (long)r * k2);
    Pb[(long)r * ld + q] = (r == q) ? var : 0.0;   // new column block (incl. the new corner)
    if (r < n_old) Pb[(long)q * ld + r] = 0.0;     // new row block
    if (r == 0) mub[q] = xy[q - n_old];
  }
}

__global__ __launch_bounds__(256) void k_fill_diag(double* __restrict__ Pb, int ld, int n,
                                                   const double* __restrict__ diag) {
  const long total = (long)n * n;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int r = (int)(e / n), c = (int)(e - (long)r * n);
    Pb[(long)r * ld + c] = (r == c) ? diag[r] : 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// launchers (called from ekf_api.hip)
// ---------------------------------------------------------------------------------------------
void launch_solve(hipStream_t st, const double* P, const double* mu_in, double* mu_out, const int* nact,
                  const StepIn* in, SolveOut* out, unsigned* flags, const DeviceConfig& cfg, int ld,
                  long pstride, int batch) {
  hipLaunchKernelGGL(k_solve, dim3(batch), dim3(64), 0, st, P, mu_in, mu_out, nact, in, out, flags, cfg,
                     ld, pstride);
}

template <int MCAP>
static void launch_panels_t(hipStream_t st, const double* P, const double* mu_in, double* mu_out,
                            const int* nact, const SolveOut* so, double* V, double* W, int ld,
                            long pstride, int batch, int n_hi) {
  hipLaunchKernelGGL(k_panels<MCAP>, dim3((n_hi + 63) / 64, batch), dim3(64), 0, st, P, mu_in, mu_out,
                     nact, so, V, W, ld, pstride);
}

void launch_panels(hipStream_t st, int mcap, const double* P, const double* mu_in, double* mu_out,
                   const int* nact, const SolveOut* so, double* V, double* W, int ld, long pstride,
                   int batch, int n_hi) {
  switch (mcap) {
    case 1: launch_panels_t<1>(st, P, mu_in, mu_out, nact, so, V, W, ld, pstride, batch, n_hi); break;
    case 2: launch_panels_t<2>(st, P, mu_in, mu_out, nact, so, V, W, ld, pstride, batch, n_hi); break;
    case 4: launch_panels_t<4>(st, P, mu_in, mu_out, nact, so, V, W, ld, pstride, batch, n_hi); break;
    case 8: launch_panels_t<8>(st, P, mu_in, mu_out, nact, so, V, W, ld, pstride, batch, n_hi); break;
    default: launch_panels_t<16>(st, P, mu_in, mu_out, nact, so, V, W, ld, pstride, batch, n_hi); break;
  }
}

template <int MCAP, int UNR, int WAVES, bool NT>
static void launch_pass_t(hipStream_t st, double* P, const double* V, const double* W,
                          const int* nact, const SolveOut* so, int ld, long pstride, int batch,
                          int n_hi, int rows_per_block) {
  dim3 grid((n_hi + 128 * WAVES - 1) / (128 * WAVES), (n_hi + rows_per_block - 1) / rows_per_block, batch);
  hipLaunchKernelGGL((k_pass<MCAP, UNR, WAVES, NT>), grid, dim3(WAVES * 64), 0, st, P, V, W, nact, so, ld,
                     pstride, rows_per_block);
}

// streaming = the batch's covariances do not fit the Infinity Cache: nontemporal, 4 waves x 8 rows;
// resident  = plain accesses, 8 waves x 4 rows (measured best on a 128 MB P, tools/pass_bench.hip).
template <int MCAP>
static void launch_pass_m(hipStream_t st, bool streaming, double* P, const double* V, const double* W,
                          const int* nact, const SolveOut* so, int ld, long pstride, int batch, int n_hi,
                          int rows_per_block) {
  constexpr int U_S = MCAP >= 16 ? 4 : 8;
  if (streaming) launch_pass_t<MCAP, U_S, 4, true>(st, P, V, W, nact, so, ld, pstride, batch, n_hi, rows_per_block);
  else launch_pass_t<MCAP, 4, 8, false>(st, P, V, W, nact, so, ld, pstride, batch, n_hi, rows_per_block);
}

void launch_pass(hipStream_t st, int mcap, bool streaming, double* P, const double* V, const double* W,
                 const int* nact, const SolveOut* so, int ld, long pstride, int batch, int n_hi,
                 int rows_per_block) {
  switch (mcap) {
    case 1: launch_pass_m<1>(st, streaming, P, V, W, nact, so, ld, pstride, batch, n_hi, rows_per_block); break;
    case 2: launch_pass_m<2>(st, streaming, P, V, W, nact, so, ld, pstride, batch, n_hi, rows_per_block); break;
    case 4: launch_pass_m<4>(st, streaming, P, V, W, nact, so, ld, pstride, batch, n_hi, rows_per_block); break;
    case 8: launch_pass_m<8>(st, streaming, P, V, W, nact, so, ld, pstride, batch, n_hi, rows_per_block); break;
    default: launch_pass_m<16>(st, streaming, P, V, W, nact, so, ld, pstride, batch, n_hi, rows_per_block); break;
  }
}

void launch_predict_rc(hipStream_t st, double* P, const double* mu_in, double* mu_out, const int* nact,
                       const SolveOut* so, int ld, long pstride, int batch, int n_hi) {
  hipLaunchKernelGGL(k_predict_rc, dim3((n_hi + 255) / 256, batch), dim3(256), 0, st, P, mu_in, mu_out,
                     nact, so, ld, pstride);
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
