// Fused cadence for uploaded streams (gfx950, wave64): all steps between two covariance passes as ONE solve launch and
// ONE panel launch.
//
// Per step the two-kernel path (ekf_kernels.hip: k_solve + k_panels) re-gathers what it needs of the CURRENT covariance:
// the base entries P_base(C, i) of that step's observed indices plus every pending rank at every state index (V and W
// re-read by every step of a cadence: 14 of k_panels' 34 us at N = 2000 x 32), and pays a launch pair and a scattered
// gather per step.  Right after a covariance pass nothing is pending, and an uploaded stream knows the landmark indices
// of its next steps.  So the union panel
//     X[a][i] = P(C_u[a], i),   C_u = {0, 1, 2} + the landmark indices of ALL steps up to the next pass (<= 83 rows)
// is gathered from P_base ONCE, and the predictions (src/replay_no_ros.py:368-430) and sequential landmark updates
// (:436-480) of all those steps are replayed on it in registers -- the same recurrences as k_panels, with every pending
// rank's effect carried in X instead of re-read from memory -- and the 80 ranks are appended once:
//   k_solve_cad   one 8-wave workgroup per trajectory: the sequential chain on the c_u x c_u block in LDS (motion
//                 models, predictions, per landmark: Jacobian at the current mean, S, K, down-date), emitting one record
//                 per landmark {H (2x5), S^-1, y, K[C_u[a], :] for the positions a later landmark still reads}
//   k_panels_cad  thread per state index i >= 3: gathers X[:, i] (83 loads in flight at once), replays, writes
//                 V[k][i], W[i][k] for all ranks, its share of the in-place prediction (rows 0, 1 of P_base) and mu[i]
// State indices 0..2 (the pose) are the solve's: it has the whole pose block, so it computes their V / W entries, the pose
// block's share of P_base's rows 0, 1 and the pose mean itself -- the panel kernel's replay has no special case for the
// first lanes.  Of these only the mean is written by the solve: the rest travels in the record and is stored by the panel
// launch (pose_epilogue), so that the solve touches neither V, W, P_base nor anything else a covariance pass uses -- which
// lets the solve of the NEXT cadence run beside the pass of this one (ekf_api.hip: look-ahead).
// Slot layout of C_u (CadGeom, ekf_device.h): later landmarks at LOWER positions, so what is still needed is always a
// prefix of the positions: compile-time bounds for the panel's register array X, a shrinking prefix of lanes here.
// Round 5: the cadence is PACKED (ekf_device.h: CadPlan): a trajectory's next <= 40 landmark updates, whatever steps they
// belong to and however many each step holds, with every prediction in between; every trajectory of the bank walks its own
// sequence.  One instantiation of each kernel serves every landmark count (rounds 3 - 4 had one per slot size 1/2/4/8/16).
// Same algebra as the per-step path in a different summation order: results agree to rounding (1e-10 relative guaranteed, 1e-13 .. 1e-12 measured,
// tests/test_gpu_cadence.py), not bit for bit.
#include <algorithm>
#include <type_traits>
#include <utility>

#include "ekf_devfn.h"

namespace ekf {

// Diagnostic build (-DCAD_STAMPS): s_memtime stamps of the solve's phases for landmark slots CAD_STAMP_S0.. (3 slots x 16
// stamps) into the record's `prow` area; tools/cad_stamps.py reads them through ekf_debug_cad.
#ifdef CAD_STAMPS
#ifndef CAD_STAMP_S0
#define CAD_STAMP_S0 16
#endif
#define CSTAMP(w, s, k)                                                                          \
  do {                                                                                           \
    if (wave == (w) && (s) >= CAD_STAMP_S0 && (s) < CAD_STAMP_S0 + 3) {                          \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      unsigned long long t_;                                                                     \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      if (lane == 0) reinterpret_cast<unsigned long long*>(&o.prow[0][0])[((s) - CAD_STAMP_S0) * 16 + (k)] = t_;   \
    }                                                                                            \
  } while (0)
#else
#define CSTAMP(w, s, k) do { } while (0)
#endif

#ifndef CAD_KS_WAVES_OVERRIDE
constexpr int CAD_KS_WAVES = 512;       // panel launches of up to this many waves of state indices take the row-split form
#else
constexpr int CAD_KS_WAVES = CAD_KS_WAVES_OVERRIDE;   // (diagnostic builds: the row-split form everywhere / nowhere)
#endif
typedef double v2d_u __attribute__((ext_vector_type(2), aligned(8)));   // two adjacent doubles, 8-byte aligned: one 16-byte load
constexpr int CAD_CS = 88;              // LDS row stride of the block (doubles): 83 columns, rows 16-byte aligned
constexpr int CAD_ROWS = 84;
constexpr int CAD_NW = 8;               // waves of the solve workgroup (512 threads: the register budget of 2 waves per SIMD;
                                        // with 16 waves the chain's constants spilled)
constexpr int CAD_DW = CAD_NW - 2;      // waves that share a down-date by rows (all but the mean wave and the record wave)
constexpr int CAD_DCH = 7;              // rows of a down-date chunk (all reads of a chunk in flight together)
constexpr int CAD_DQ = (CAD_CU - 2 + CAD_DCH * CAD_DW - 1) / (CAD_DCH * CAD_DW) * CAD_DCH;   // rows per down-date wave

constexpr int CHAIN_SPIN_LIMIT = 1 << 20;   // bounded device-scope waits: x (s_sleep + one load past the L2): about a second
// ---- device-scope hand-overs between kernels that run side by side (the two streams of a chained run) ----
// The XCDs' L2s are not coherent with one another, and a fence that makes them so writes back / invalidates a whole L2
// (ekf_kernels.hip: mailbox_publish).  So what crosses between kernels in flight is written THROUGH (device-scope relaxed
// stores: sc1), waited for (vmcnt), and announced by a counter written the same way; the reader polls the counter and reads
// with device-scope loads (past its own L2).  Counters only ever grow; a wait is for "at least `target`" in wrap-around
// arithmetic, bounded (a timeout raises EKF_FLAG_INTERNAL: the state is undefined from there on, like k_step_split's).
constexpr int SYNC_STRIDE = 32;         // words between two counters (a cache line each)
enum { SYNC_SOLVE = 0, SYNC_PASS = 1, SYNC_START = 2, SYNC_GATHER = 3, SYNC_WORDS = 4 * SYNC_STRIDE };
__device__ __forceinline__ double ld_dev(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool sync_wait(const unsigned* word, unsigned target) {
  for (int spin = 0; spin < CHAIN_SPIN_LIMIT; ++spin) {
    if ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) return true;
    __builtin_amdgcn_s_sleep(8);
  }
  return false;
}

// The positions C_u of a trajectory's cadence, formed the same way by the solve and by the look-ahead gather (512 threads
// each).  Thread p < CAD_SLOTS takes touched step p of the plan: its landmark range [lo, hi) and, through a 40-entry prefix sum
// in LDS, the slots in front of it.  The landmarks themselves are dealt over all threads -- entry e = (step e / 16, landmark
// e % 16), up to 640 of them -- and FETCHED BEFORE the counts are known (a record's 16 landmark places always exist): one
// memory round trip for the whole prologue instead of two dependent ones.  `per_slot(s, range, bearing)` lets the solve pick
// up the measurements (WANT_Z).  Returns (to every thread) the number of slots.  cntS / firstS / loS: CAD_SLOTS + 1 ints each.
template <bool WANT_Z, class PerSlot>
__device__ __forceinline__ int cad_positions(const CadPlan& pl, const StepIn* __restrict__ in, int batch, int b, const DeviceConfig& cfg,
                                             int tid, int* Cs, int* cntS, int* firstS, int* loS, PerSlot per_slot) {
  using G = CadGeom;
  constexpr int NT = 64 * CAD_NW, EPT = (CAD_SLOTS * MMAX + NT - 1) / NT;   // entries per thread (2)
  if (tid < 128) Cs[tid] = tid < 3 ? tid : 0;
  int eidx[EPT];
  double er[EPT], eb[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = tid + NT * k, p = e / MMAX, q = e - p * MMAX;
    eidx[k] = 0;
    er[k] = 0.0;
    eb[k] = 0.0;
    if (p < pl.ns) {
      const StepIn* st = in + ((long)(pl.t0 + p) * batch + b);
      eidx[k] = st->idx[q];
      if (WANT_Z) {
        er[k] = st->range[q];
        eb[k] = st->bearing[q];
      }
    }
  }
  int cnt = 0, lo = 0;
  if (tid < pl.ns) {
    const StepIn* st = in + ((long)(pl.t0 + tid) * batch + b);
    const int m = ((st->flags & FLAG_UPDATE) && cfg.enable_measurement_model) ? min(st->m, MMAX) : 0;
    lo = tid == 0 ? pl.j0 : 0;
    const int hi = tid == pl.ns - 1 ? min(pl.jend, m) : m;
    cnt = max(hi - lo, 0);
  }
  if (tid <= CAD_SLOTS) {
    cntS[tid] = tid < CAD_SLOTS ? cnt : 0;
    loS[tid] = lo;
  }
  __syncthreads();
  if (tid <= CAD_SLOTS) {
    int f = 0;
    for (int u = 0; u < tid; ++u) f += cntS[u];
    firstS[tid] = f;
  }
  __syncthreads();
  const int nslots = min(firstS[CAD_SLOTS], CAD_SLOTS);
  const int s0 = G::GM - nslots;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = tid + NT * k, p = e / MMAX, q = e - p * MMAX;
    if (p < pl.ns) {
      const int lo_p = loS[p], j = q - lo_p;
      if (j >= 0 && j < cntS[p]) {
        const int sl = s0 + firstS[p] + j;
        if (sl < G::GM) {                              // (a plan that disagrees with the records cannot write outside)
          const int pq = G::pa(sl);
          Cs[pq] = 3 + 2 * eidx[k];
          Cs[pq + 1] = 4 + 2 * eidx[k];
          per_slot(sl, er[k], eb[k]);
        }
      }
    }
  }
  return nslots;
}

// CHAIN (round 6, chained solves): the instantiation of a run whose solves follow one another on the handle's stream
// (ekf_api.hip: enqueue_cadence).  It also down-dates the pose block behind the LAST landmark and records it
// (CadOut::posefin): with the per-landmark records that is all k_chain_cad needs to form the next cadence's block without
// this cadence's panel launch and covariance pass.  `gmu` (with gbuf): the mean at the cadence's positions, from k_chain_cad.
template <bool CHAIN>
__global__ __launch_bounds__(64 * CAD_NW) void k_solve_cad(const double* __restrict__ P,
                                                    const double* __restrict__ mu_in, double* __restrict__ mu_out,
                                                    double* __restrict__ dacc_out, const int* __restrict__ nact,
                                                    const StepIn* __restrict__ in, const CadPlan* __restrict__ plan, int batch,
                                                    CadOut* __restrict__ out, unsigned* __restrict__ flags,
                                                    DeviceConfig cfg, int ld,
                                                    long pstride, const double* __restrict__ gbuf, int gparts,
                                                    double* __restrict__ colbuf, int col_wgs, int n_hi,
                                                    const double* __restrict__ gmu, unsigned* __restrict__ sync,
                                                    unsigned start_sigma, const CadPre* __restrict__ pre) {
  using G = CadGeom;
  constexpr int GM = G::GM, CU = G::CU;
  __shared__ __attribute__((aligned(16))) double Pc[CAD_ROWS][CAD_CS];
  __shared__ double2 hpS[128], kcS[128];
  __shared__ int Cs[128];
  __shared__ double2 zS[CAD_SLOTS];                    // (range, bearing) of slot s
  __shared__ double2 laS[CAD_SLOTS];                   // (lin, ang) of touched step p
  __shared__ int mS[CAD_SLOTS + 1], firstS[CAD_SLOTS + 1], loS[CAD_SLOTS + 1], fS[CAD_SLOTS];
  __shared__ double mot[4];                            // G[0,2], G[1,2] of the step being predicted
  __shared__ double2 hS[2][6];                         // linearisation of slot s in hS[s & 1]: {h[0][k], h[1][k]}, k < 5
  __shared__ double2 siS[2];                           // S^-1 of the slot in flight
  __shared__ double2 yS[2];                            // innovation of slot s in yS[s & 1] (for its record)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)blockIdx.x >= batch) {
    // ---- the COLUMN GATHER beside the chain (round 5) ----
    // The chain keeps one CU per trajectory busy for 55 - 70 us and the rest of the chip idle.  What the panel launch behind it
    // gathers of the covariance has two halves: P(C_u[a], i) for i >= C_u[a] lies in row C_u[a] (64 state indices: 512
    // contiguous bytes), for i < C_u[a] it is stored mirrored, as P_base(i, C_u[a]): one 16-byte pair per ROW i, a 64-byte
    // sector of its own each, and random sectors stream at a third of the rate rows do (PMC: 107 MB more fetched for 62 us
    // more at N = 2000 x 32 when the landmarks of a cadence are scattered over the map instead of consecutive,
    // profiles/r05_scattered_indices.txt).  Nothing of that depends on the chain: workgroups batch.. of THIS launch fetch
    // the mirrored pairs meanwhile -- 512 state indices each, every wave the pairs that lie entirely beyond its 64 indices --
    // and lay them down coalesced, colbuf[b][a][i]; the panel launch reads them as rows.  (The positions C_u are formed
    // from the plan exactly as the chain forms them.)
    // `col_wgs` workgroups (one per CU the chain leaves free, so that all of them are resident at once) share the items
    // (trajectory, 64 state indices), a wave at a time, all of a wave's loads in flight together.  An item costs what lies
    // beyond its state indices -- everything for the first strip, nothing for the last -- and there are a few more items than
    // waves (N = 2000 x 32: 2016 for 1792): dealt strip-major, dearest first, so that the second round is the cheap tail.
    // (Tickets from a global counter instead: 80 us against 60 -- 3800 atomics on one word; profiles/r05_scattered_indices.txt.)
    __shared__ int CsG[CAD_NW][128];                   // (per wave: the positions of the trajectory its item belongs to)
    __shared__ int cntG[CAD_NW][CAD_SLOTS + 1];
    const int strips = (n_hi + 63) >> 6, items = batch * strips;
    const int gwave = ((int)blockIdx.x - batch) * CAD_NW + wave;
    for (int item = gwave; item < items; item += col_wgs * CAD_NW) {
      const int strip = item / batch, b = item - strip * batch, i0 = 64 * strip, i = i0 + lane;
      const CadPlan pl = plan[b];
      const int n = nact[b];
      if (i0 >= n || i0 >= min(n, pl.neff) || pl.nslots == 0) continue;   // (uniform) nothing of this wave is replayed
      // the positions C_u of trajectory b, by this wave alone (cad_positions restated for one wave: lane p = touched step p)
      int* Cw = CsG[wave];
      int* cw = cntG[wave];
      Cw[lane] = lane < 3 ? lane : 0;
      Cw[64 + lane] = 0;
      int cnt = 0, lo = 0;
      const StepIn* st = nullptr;
      if (lane < pl.ns) {
        st = in + ((long)(pl.t0 + lane) * batch + b);
        const int m = ((st->flags & FLAG_UPDATE) && cfg.enable_measurement_model) ? min(st->m, MMAX) : 0;
        lo = lane == 0 ? pl.j0 : 0;
        const int hi = lane == pl.ns - 1 ? min(pl.jend, m) : m;
        cnt = max(hi - lo, 0);
      }
      if (lane <= CAD_SLOTS) cw[lane] = lane < CAD_SLOTS ? cnt : 0;
      WAVE_LDS_SYNC();
      int first = 0;
      for (int u = 0; u < lane && u < CAD_SLOTS; ++u) first += cw[u];
      const int nslots = min(pl.nslots, CAD_SLOTS), s0 = GM - nslots;
      if (lane < pl.ns) {
        for (int j = 0; j < cnt; ++j) {
          const int sl = s0 + first + j;
          if (sl < GM) {
            const int idx = st->idx[lo + j], pq = G::pa(sl);
            Cw[pq] = 3 + 2 * idx;
            Cw[pq + 1] = 4 + 2 * idx;
          }
        }
      }
      WAVE_LDS_SYNC();
      const int ii = i < n ? i : n - 1;
      const double* Pb = P + (long)b * pstride;
      double* cb = colbuf + ((long)b * CAD_CU) * ld;
      v2d_u v[CAD_SLOTS];
      unsigned long long taken = 0ull;                 // (uniform) bit q: pair q is mirrored for this whole wave
      // Pairs whose neighbours in the cadence lie in the same 128-byte line of the row (consecutive landmarks: the 16 columns
      // of a step share one) want the caches -- eight pairs per line fetched once; a pair alone in its line should stream past
      // them (nontemporal: scattered landmarks 106 -> 95 us for the launch; clustered ones lose 8 us when they stream).
      // Decided per ITEM -- one branch around two copies of the loop: a choice per load merges 40 times and the loads wait
      // for one another (profiles/r05_scattered_indices.txt).
      int clustered = 0;
      if (lane < nslots) {
        const int a = 3 + 2 * lane, c0 = Cw[a];
        const int cprev = lane > 0 ? Cw[a - 2] : -64, cnext = lane + 1 < nslots ? Cw[a + 2] : -64;
        clustered = (abs(c0 - cprev) < 16 || abs(c0 - cnext) < 16) ? 1 : 0;
      }
      const bool stream_past = 2 * __popcll(__ballot(clustered != 0)) < nslots;   // (uniform) most pairs are alone in their lines
#define EKF_COLG_LOADS(LOAD)                                                                                              \
  _Pragma("unroll") for (int q = 0; q < CAD_SLOTS; ++q) {                                                                 \
    const int a = 3 + 2 * q;                                                                                              \
    const int c0 = Cw[a], c1 = Cw[a + 1];                                                                                 \
    const bool take = q < nslots && c1 == c0 + 1 && i0 + 63 <= c0 && (c1 & (PPW - 1)) != 0; /* (uniform: k_panels_cad's condition) */ \
    v[q].x = 0.0;                                                                                                         \
    v[q].y = 0.0;                                                                                                         \
    if (take) {                                                                                                           \
      v[q] = LOAD(reinterpret_cast<const v2d_u*>(Pb + p_index(ld, ii, c0)));                                              \
      taken |= 1ull << q;                                                                                                 \
    }                                                                                                                     \
  }
#define EKF_LOAD_CACHED(p) (*(p))
#define EKF_LOAD_STREAM(p) __builtin_nontemporal_load(p)
      if (stream_past) {
        EKF_COLG_LOADS(EKF_LOAD_STREAM)
      } else {
        EKF_COLG_LOADS(EKF_LOAD_CACHED)
      }
#undef EKF_LOAD_STREAM
#undef EKF_LOAD_CACHED
#undef EKF_COLG_LOADS
#pragma unroll
      for (int q = 0; q < CAD_SLOTS; ++q) {
        if (((taken >> q) & 1ull) && i < n) {
          const int a = 3 + 2 * q;
          cb[(long)a * ld + i] = v[q].x;
          cb[(long)(a + 1) * ld + i] = v[q].y;
        }
      }
      WAVE_LDS_SYNC();                                 // (the next item rewrites this wave's positions)
    }
    return;
  }
  const int b = blockIdx.x;
  const double* Pb = P + (long)b * pstride;
  const double* mu_in_b = mu_in + (long)b * ld;
  CadOut& o = out[b];
  const CadPlan pl = plan[b];
  const int nsteps = pl.ns;                            // touched steps
  // (chained) this workgroup is placed: the previous cadence's covariance pass may fill the rest of the chip now (the panel launch
  // in front of it waits for this word -- a pass that got there first keeps every CU busy for its whole duration, and the solve,
  // which needs a CU to itself, 20 us from being placed: profiles/r06_chained_solves.txt)
  if constexpr (CHAIN) {
    if (sync && threadIdx.x == 0) __hip_atomic_store(sync + SYNC_START * SYNC_STRIDE, start_sigma, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // (chained) block and mean come from k_chain_cad, whole (84 x 88, zeros beyond the positions in use): fetched before the
  // positions are formed -- nothing of it depends on them -- so that the two round trips overlap (2 us of the launch)
  constexpr int RQP = (CAD_CU + CAD_NW - 1) / CAD_NW;  // rows per wave
  double pre0[CHAIN ? RQP : 1], pre1[CHAIN ? RQP : 1], pmu0 = 0.0, pmu1 = 0.0;
  if constexpr (CHAIN) {
    if (gmu) {                                         // (uniform)
#pragma unroll
      for (int q = 0; q < RQP; ++q) {
        const double* gb = gbuf + ((long)b * CAD_ROWS + min(wave + CAD_NW * q, CAD_ROWS - 1)) * CAD_CS;
        pre0[q] = gb[lane];
        pre1[q] = gb[min(64 + lane, CAD_CS - 1)];
      }
      if (wave == 1) {
        pmu0 = gmu[(long)b * 128 + lane];
        pmu1 = gmu[(long)b * 128 + 64 + lane];
      }
    }
  }

  // ---- inputs: the plan's steps (thread p: touched step p), their landmarks' slots and positions ----
  int nslots;
  if (CHAIN && pre) {
    // (uniform) chained: formed one cadence ahead by the chain launch's positions workgroup (CadPre) -- one coalesced round trip
    const CadPre& pp = pre[b];
    if (tid < 128) Cs[tid] = pp.C[tid];
    if (tid <= CAD_SLOTS) {
      mS[tid] = pp.cnt[tid];
      firstS[tid] = pp.first[tid];
      loS[tid] = pp.lo[tid];
    }
    if (tid < CAD_SLOTS) {
      fS[tid] = pp.fl[tid];
      laS[tid] = make_double2(pp.la[tid][0], pp.la[tid][1]);
      zS[tid] = make_double2(pp.z[tid][0], pp.z[tid][1]);
    }
    nslots = pp.nslots;
  } else {
    if (tid < CAD_SLOTS) {
      int fl = 0;
      double2 la = make_double2(0.0, 0.0);
      if (tid < nsteps) {
        const StepIn& st = in[(long)(pl.t0 + tid) * batch + b];
        fl = st.flags;
        if (tid == 0 && pl.j0 > 0) fl &= ~FLAG_PREDICT;   // a step cut by the previous cadence: its prediction has happened
        la = make_double2(st.lin, st.ang);
      }
      fS[tid] = fl;
      laS[tid] = la;
    }
    nslots = cad_positions<true>(pl, in, batch, b, cfg, tid, Cs, mS, firstS, loS,
                                 [&](int s, double zr, double zb) { zS[s] = make_double2(zr, zb); });
  }
  const int s0 = GM - nslots;
  const int cu = 3 + 2 * nslots;                       // positions in use
  const int neff_eff = min(nact[b], pl.neff);
  __syncthreads();
  const int Cl0 = Cs[lane], Cl1 = Cs[64 + lane];       // positions lane and 64 + lane
  if (tid <= CU) o.C[tid] = tid < CU ? Cs[tid] : 0;
  if (tid < CAD_SLOTS) {
    // slot of touched step p's first landmark -- or of the next step's that has one: the panel launch applies a step's
    // prediction when it reaches that slot
    o.sfirst[tid] = tid < nsteps ? s0 + firstS[tid] : GM;
  }
  if (tid == 0) {
    o.nslots = nslots;
    o.neff = neff_eff;
    o.npred = nsteps;
    o.pad0 = 0;
  }

  // ---- the mean wave (wave 1): lane l holds the mean at positions l and 64 + l ----
  double mu0 = 0.0, mu1 = 0.0, y0 = 0.0, y1 = 0.0;
  double rdsum0 = 0.0, rdsum1 = 0.0, rdsum2 = 0.0;     // (wave 1) pose-block noise of the whole cadence
  LinGeom lg{};
  auto mean_at = [&](int p) -> double {                // p wave-uniform
    return p < 64 ? read_lane(mu0, p) : read_lane(mu1, p - 64);
  };
  // motion model of step t (src/replay_no_ros.py:368-417) at the current pose mean; publishes G[0,2], G[1,2]
  auto motion = [&](int t) {
    const double2 la = laS[t];
    const bool do_pred = (fS[t] & FLAG_PREDICT) != 0;
    const double th = read_lane(mu0, 2);
    double g0 = 0.0, g1 = 0.0, nx = read_lane(mu0, 0), ny = read_lane(mu0, 1), nth = th;
    if (do_pred && !cfg.disable_motion_model) {
      const double lin = la.x, ang = la.y;
      double s0, c0;
      sincos(th, &s0, &c0);
      if (cfg.enable_circular_interpolation && fabs(ang) > cfg.arc_threshold) {   // :390 arc
        double s1, c1;
        sincos(th + ang, &s1, &c1);
        const double r = lin / ang;
        nx += -r * s0 + r * s1;
        ny += r * c0 - r * c1;
        nth = wrap_pi(th + ang);                       // :397
        g0 = -r * c0 + r * c1;                         // :401
        g1 = -r * s0 + r * s1;                         // :402
      } else {                                         // :376 straight / :405-417 linear mode
        nx += lin * c0;
        ny += lin * s0;
        if (!cfg.enable_circular_interpolation) nth = th + ang;   // no wrap (:409); :381 keeps theta
        g0 = -lin * s0;
        g1 = lin * c0;
      }
    }
    if (lane < 3) mu0 = lane == 0 ? nx : (lane == 1 ? ny : nth);
    if (do_pred) {
      rdsum0 += cfg.rd[0];
      rdsum1 += cfg.rd[1];
      rdsum2 += cfg.rd[2];
    }
    if (lane == 0) {
      mot[0] = g0;
      mot[1] = g1;
      *reinterpret_cast<double2*>(o.g[t]) = make_double2(g0, g1);
    }
  };
  auto jacobian_at_mean = [&](int p, int par) {        // landmark at positions p, p + 1: publishes hS[par], keeps the geometry
    double hn[2][5];
    lg = linearize_h(read_lane(mu0, 0), read_lane(mu0, 1), read_lane(mu0, 2), mean_at(p), mean_at(p + 1), hn);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 5; ++k) hS[par][k] = make_double2(hn[0][k], hn[1][k]);
    }
  };

  // ---- gather the block P[C_u, C_u] (nothing is pending: P = P_base -- or the look-ahead gather's copy); wave 1 starts on
  // its means meanwhile ----
  {
    constexpr int RQ = (CU + CAD_NW - 1) / CAD_NW;     // rows per wave
    const int lane_b = min(64 + lane, CAD_CS - 1);
    double gv0[RQ], gv1[RQ];
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      gv0[q] = 0.0;
      gv1[q] = 0.0;
    }
    if (CHAIN && gmu) {                                // (uniform) chained: fetched at the top of the launch
      if constexpr (CHAIN) {
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
          gv0[q] = pre0[q];
          gv1[q] = pre1[q];
        }
      }
    } else if (gbuf) {
      // (uniform) look-ahead: the block was gathered (base + the ranks still pending then) by k_gather_cad, in `gparts`
      // parts, added here in a fixed order; all loads of a part are in flight together
      // (... four parts' loads in flight together: ten dependent round trips were 8 us of a single trajectory's cadence)
      constexpr int GPB = 4;
#pragma unroll
      for (int g0 = 0; g0 < KTOT / 8; g0 += GPB) {
        if (g0 < gparts) {                             // (uniform)
          double t0[GPB][RQ], t1[GPB][RQ];
#pragma unroll
          for (int u = 0; u < GPB; ++u) {
            const bool on = g0 + u < gparts && g0 + u < KTOT / 8;   // (uniform)
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
              const int r = min(wave + CAD_NW * q, max(cu - 1, 0));
              const double* gb = gbuf + (((long)(on ? g0 + u : 0) * batch + b) * CAD_ROWS + r) * CAD_CS;
              t0[u][q] = on ? gb[lane] : 0.0;
              t1[u][q] = on ? gb[lane_b] : 0.0;
            }
          }
#pragma unroll
          for (int u = 0; u < GPB; ++u) {
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
              gv0[q] += t0[u][q];
              gv1[q] += t1[u][q];
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < RQ; ++q) {
        const int r = wave + CAD_NW * q;
        if (r < cu) {                                  // (wave-uniform)
          const int Cr = Cs[r];
          gv0[q] = Pb[p_index(ld, min(Cr, Cl0), max(Cr, Cl0))];     // the upper triangle is authoritative
          if (cu > 64) gv1[q] = Pb[p_index(ld, min(Cr, Cl1), max(Cr, Cl1))];
        }
      }
    }
    if (wave == 1) {
      if (CHAIN && gmu) {                              // (uniform) chained: the mean at the positions, from k_chain_cad
        mu0 = pmu0;
        mu1 = pmu1;
      } else {
        mu0 = mu_in_b[Cl0];
        mu1 = mu_in_b[Cl1];
      }
    }
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      const int r = wave + CAD_NW * q;
      if (r < cu) {
        Pc[r][lane] = gv0[q];
        if (64 + lane < CAD_CS) Pc[r][64 + lane] = gv1[q];
      }
    }
  }
  // wave roles: 0 = the covariance chain (+ a down-date share), 1 = the mean, CAD_NW - 1 = the records (everything the
  // panel kernel gets goes to memory from there, off the chain), the others: down-date
  const bool rec_wave = wave == CAD_NW - 1;
  const int ds = wave == 0 ? 0 : wave - 1;             // down-date slot of this wave (waves 0, 2 .. CAD_NW - 2)
  if (wave == 1) motion(0);
  WG_LDS_BARRIER();
#ifndef CAD_STAMPS
  if (wave == 0) {                                     // (diagnostic record) rows 0, 1 of the block before the cadence
    if (lane < CU) {
      o.prow[0][lane] = Pc[0][lane];
      o.prow[1][lane] = Pc[1][lane];
    }
    if (64 + lane <= CU) {
      o.prow[0][64 + lane] = 64 + lane < CU ? Pc[0][64 + lane] : 0.0;
      o.prow[1][64 + lane] = 64 + lane < CU ? Pc[1][64 + lane] : 0.0;
    }
  }
#endif

  double dd0 = 0.0, dd1 = 0.0;                         // (wave 0, lanes 0..2) in-place change of P_base(0, l), P_base(1, l)
  for (int t = 0; t < nsteps; ++t) {
    const int m = __builtin_amdgcn_readfirstlane(mS[t]);
    const int s_first = s0 + __builtin_amdgcn_readfirstlane(firstS[t]);   // (a step without landmarks: the next step's first slot)
    const int ca = G::pa(s_first) + 2;                 // positions in use from this step on: [0, ca)  (s_first == GM: the pose)
    const bool two = ca > 64;                          // (uniform) the second half of the columns is live
    // ---- prediction of step t on the block: P' = G P G^T + R restricted to C_u (:428-430).  Only rows / columns 0, 1
    // change, and the block is exactly symmetric: lane r holds P[0..2][r] = P[r][0..2] and produces P'[r][0], P'[r][1],
    // which for r >= 2 are also P'[0][r], P'[1][r].  Wave 1 linearises the step's first landmark meanwhile.
    if (wave == 0) {
      const bool do_pred = (fS[t] & FLAG_PREDICT) != 0;
      const double g0 = mot[0], g1 = mot[1];
      const double rd0 = do_pred ? cfg.rd[0] : 0.0, rd1 = do_pred ? cfg.rd[1] : 0.0, rd2 = do_pred ? cfg.rd[2] : 0.0;
      const double s20 = Pc[2][0], s21 = Pc[2][1], p22 = Pc[2][2];
      const int r0 = min(lane, ca - 1), r1 = min(64 + lane, CAD_CS - 1);
      const double p0 = Pc[0][r0], p1 = Pc[1][r0], p2 = Pc[2][r0];
      const double q0 = Pc[0][r1], q1 = Pc[1][r1], q2 = Pc[2][r1];
      const double gr = lane == 0 ? g0 : (lane == 1 ? g1 : 0.0);
      double x0 = p0, x1 = p1, x2 = p2;                // row r of G P, columns 0..2 (rows 0, 1 take g_r x row 2)
      if (lane < 2) {
        x0 = fma(gr, s20, p0);
        x1 = fma(gr, s21, p1);
        x2 = fma(gr, p22, p2);
      }
      double c0n = fma(g0, x2, x0), c1n = fma(g1, x2, x1);
      dd0 += fma(g0, x2, lane < 2 ? gr * s20 : 0.0);   // P'(0, l) - P(0, l) and P'(1, l) - P(1, l) without the noise
      dd1 += fma(g1, x2, lane < 2 ? gr * s21 : 0.0);
      if (lane == 0) c0n += rd0;
      if (lane == 1) c1n += rd1;
      const double e0n = fma(g0, q2, q0), e1n = fma(g1, q2, q1);
      if (lane < ca) {
        Pc[lane][0] = c0n;
        Pc[lane][1] = c1n;
        if (lane >= 2) {
          Pc[0][lane] = c0n;
          Pc[1][lane] = c1n;
        }
        if (lane == 2) Pc[2][2] = p2 + rd2;
      }
      if (two && 64 + lane < ca) {
        Pc[64 + lane][0] = e0n;
        Pc[64 + lane][1] = e1n;
        Pc[0][64 + lane] = e0n;
        Pc[1][64 + lane] = e1n;
      }
    } else if (wave == 1) {
      if (m > 0) jacobian_at_mean(G::pa(s_first), s_first & 1);
    }
    WG_LDS_BARRIER();                                  // S0(t): predicted block and the first Jacobian published
    if (wave == 1 && m > 0) {
      const double2 z = zS[s_first];
      innovation(lg, z.x, z.y, y0, y1);
      if (lane == 0) yS[s_first & 1] = make_double2(y0, y1);
    }
    // ---- the step's landmarks, sequentially (:436-480) ----
    for (int j = 0; j < m; ++j) {
      const int s = s_first + j, pa = G::pa(s);        // this landmark sits at positions pa, pa + 1; [0, pa) lives on
      const bool two_j = pa + 2 > 64;                  // columns 64.. still in use
      const bool last = j + 1 == m && t + 1 == nsteps; // nothing reads the block after this landmark
      double2 hpa = make_double2(0.0, 0.0), hpb = make_double2(0.0, 0.0);   // (H P)[:, l] of this wave's columns
      CSTAMP(0, s, 0);
      CSTAMP(1, s, 8);
      if (wave == 0) {
        // phase A: rows sel = {0, 1, 2, pa, pa + 1} of P at column l give (H P)[:, l]; P is symmetric, so P H^T is the
        // transpose and the gain needs no second product
        // (the landmark's linearisation -- published by the mean wave before the barrier -- is read with the rows: one LDS
        //  round trip for both)
        const int la = min(lane, pa + 1), lb = min(64 + lane, CAD_CS - 1);
        double h[2][5];
        double pra[5], prb[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const double2 tt = hS[s & 1][k];
          h[0][k] = tt.x;
          h[1][k] = tt.y;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const int r = k < 3 ? k : pa + (k - 3);
          pra[k] = Pc[r][la];
          prb[k] = 0.0;
        }
        if (two_j) {                                   // (uniform)
#pragma unroll
          for (int k = 0; k < 5; ++k) prb[k] = Pc[k < 3 ? k : pa + (k - 3)][lb];
        }
        hpa = make_double2(h[0][0] * pra[0], h[1][0] * pra[0]);
#pragma unroll
        for (int k = 1; k < 5; ++k) {
          hpa.x = fma(h[0][k], pra[k], hpa.x);
          hpa.y = fma(h[1][k], pra[k], hpa.y);
        }
        hpS[lane] = hpa;
        if (two_j) {
          hpb = make_double2(h[0][0] * prb[0], h[1][0] * prb[0]);
#pragma unroll
          for (int k = 1; k < 5; ++k) {
            hpb.x = fma(h[0][k], prb[k], hpb.x);
            hpb.y = fma(h[1][k], prb[k], hpb.y);
          }
          hpS[64 + lane] = hpb;
        }
        WAVE_LDS_SYNC();
        CSTAMP(0, s, 1);
        // phase B: S = H P H^T + Q (:473) from the five pairs at sel, every lane redundantly
        double2 hv[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) hv[k] = hpS[k < 3 ? k : pa + (k - 3)];
        double S00 = cfg.qd[0], S01 = 0.0, S10 = 0.0, S11 = cfg.qd[1];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          S00 = fma(hv[k].x, h[0][k], S00);
          S01 = fma(hv[k].x, h[1][k], S01);
          S10 = fma(hv[k].y, h[0][k], S10);
          S11 = fma(hv[k].y, h[1][k], S11);
        }
        const double rdet = fast_recip(S00 * S11 - S01 * S10);   // (<= 1 ulp; the IEEE division is ~250 dependent cycles of this chain)
        const double i00 = S11 * rdet, i01 = -S01 * rdet, i10 = -S10 * rdet, i11 = S00 * rdet;
        const double2 ka = make_double2(hpa.x * i00 + hpa.y * i10, hpa.x * i01 + hpa.y * i11);   // K[C_u[l], :]
        kcS[lane] = ka;
        if (two_j) kcS[64 + lane] = make_double2(hpb.x * i00 + hpb.y * i10, hpb.x * i01 + hpb.y * i11);
        if (lane == 0) {
          siS[0] = make_double2(i00, i01);
          siS[1] = make_double2(i10, i11);
        }
        CSTAMP(0, s, 2);
      }
      CSTAMP(0, s, 3);
      CSTAMP(1, s, 9);
      WG_LDS_BARRIER();                                // b1: K, (H P) and S^-1 of this landmark are in LDS
      CSTAMP(0, s, 4);
      CSTAMP(1, s, 10);
      if (wave == 1) {
        // the mean (:476); then the next landmark's Jacobian at the new mean, or the next step's motion model
        const double2 k0 = kcS[lane], k1 = kcS[64 + lane];
        if (lane < pa + 2) mu0 += k0.x * y0 + k0.y * y1;
        if (64 + lane < pa + 2) mu1 += k1.x * y0 + k1.y * y1;
        CSTAMP(1, s, 13);
#ifndef CADS_SKIP_JAC                                    /* diagnostic build: no re-linearisation (wrong results) */
        if (j + 1 < m) jacobian_at_mean(pa - 2, (s + 1) & 1);
        else if (t + 1 < nsteps) motion(t + 1);
#endif
      } else if (rec_wave) {
        // the record of this landmark for the panel kernel, and the pose's own entries of the new ranks
        double2* rec2 = reinterpret_cast<double2*>(o.rec + G::rec_off(s));
        const double2 ka = kcS[lane], kb = kcS[64 + lane];
        if (lane < pa) rec2[8 + lane] = ka;
        if (two_j && 64 + lane < pa) rec2[8 + 64 + lane] = kb;
        if (lane < 5) rec2[lane] = hS[s & 1][lane];
        if (lane == 5 || lane == 6) rec2[lane] = siS[lane - 5];
        if (lane == 7) rec2[7] = yS[s & 1];
        if (lane < 3) {
          const double2 hp = hpS[lane];
          double2* vw = reinterpret_cast<double2*>(o.posevw[s][lane]);
          vw[0] = hp;
          vw[1] = make_double2(-ka.x, -ka.y);
        }
#ifdef CADS_SKIP_DD                                     /* diagnostic build: the block is never down-dated (wrong results) */
      } else if (false) {
#else
      } else if (CHAIN || !last) {                     // (CHAIN: the last landmark too -- the pose block behind it is a result)
#endif
        // down-date (:480) of what lives on: P[r][l] -= K[r, :] . (H P)[:, l] for r, l < pa; rows ds, ds + CAD_DW, ... are
        // this wave's.  Every access is unconditional and every address one base plus a compile-time offset: a row or a
        // column >= pa is dead (nothing reads it again), so what lands there does not matter, and rows up to
        // ds + CAD_DW (CAD_DQ - 1) <= 83 exist; all reads of a chunk are in flight before its first FMA.
        static_assert(CAD_DW - 1 + CAD_DW * (CAD_DQ - 1) < CAD_ROWS, "down-date rows stay inside the block");
        if (wave != 0) {
          hpa = hpS[lane];
          if (two_j) hpb = hpS[64 + lane];
        }
        const bool lane_b = 64 + lane < CAD_CS;        // second column half: columns 64 .. CAD_CS - 1 exist
        const int lb = lane_b ? 64 + lane : 64;
#pragma unroll
        for (int q0 = 0; q0 < CAD_DQ; q0 += CAD_DCH) {
          if (ds + CAD_DW * q0 < pa) {                 // (uniform)
            double2 kr[CAD_DCH];
            double pv[CAD_DCH], pw[CAD_DCH];
#pragma unroll
            for (int u = 0; u < CAD_DCH; ++u) {
              kr[u] = kcS[ds + CAD_DW * (q0 + u)];
              pv[u] = Pc[ds + CAD_DW * (q0 + u)][lane];
              pw[u] = 0.0;
            }
            if (two_j) {                               // (uniform)
#pragma unroll
              for (int u = 0; u < CAD_DCH; ++u) pw[u] = Pc[ds + CAD_DW * (q0 + u)][lb];
            }
#pragma unroll
            for (int u = 0; u < CAD_DCH; ++u) pv[u] = fma(-kr[u].x, hpa.x, pv[u]);
#pragma unroll
            for (int u = 0; u < CAD_DCH; ++u) pv[u] = fma(-kr[u].y, hpa.y, pv[u]);
#pragma unroll
            for (int u = 0; u < CAD_DCH; ++u) Pc[ds + CAD_DW * (q0 + u)][lane] = pv[u];
            if (two_j) {
#pragma unroll
              for (int u = 0; u < CAD_DCH; ++u) pw[u] = fma(-kr[u].x, hpb.x, pw[u]);
#pragma unroll
              for (int u = 0; u < CAD_DCH; ++u) pw[u] = fma(-kr[u].y, hpb.y, pw[u]);
              if (lane_b) {
#pragma unroll
                for (int u = 0; u < CAD_DCH; ++u) Pc[ds + CAD_DW * (q0 + u)][64 + lane] = pw[u];
              }
            }
          }
        }
      }
      CSTAMP(0, s, 5);
      CSTAMP(1, s, 11);
      WG_LDS_BARRIER();                                // b2: block down-dated; next Jacobian (or the next step's G) published
      CSTAMP(0, s, 6);
      if (wave == 1 && j + 1 < m) {
        const double2 z = zS[s + 1];
        innovation(lg, z.x, z.y, y0, y1);
        if (lane == 0) yS[(s + 1) & 1] = make_double2(y0, y1);
      }
      CSTAMP(1, s, 12);
      CSTAMP(0, s, 7);
    }
    if (m == 0) {                                      // (uniform) no landmark whose tail could carry the next motion model
      if (wave == 1 && t + 1 < nsteps) motion(t + 1);
      WG_LDS_BARRIER();
    }
  }

  // ---- results the solve owns: the pose mean, the pose block of P_base's rows 0, 1, the pending pose noise ----
  if (wave == 1) {
    double* mu_out_b = mu_out + (long)b * ld;
    bool bad = false;
    if (lane < 3) mu_out_b[lane] = mu0;
    bad = !(fabs(mu0) <= 1.79769313486231570815e308) || (64 + lane < CU && !(fabs(mu1) <= 1.79769313486231570815e308));
    if (__any(bad) && lane == 0) atomicOr(flags + b, EKF_FLAG_NONFINITE);
    if (lane == 0) {
      dacc_out[4 * b + 0] = rdsum0;
      dacc_out[4 * b + 1] = rdsum1;
      dacc_out[4 * b + 2] = rdsum2;
      o.rdsum[0] = rdsum0;                             // (for a cadence that appends no rank anywhere in the bank: see pose_epilogue)
      o.rdsum[1] = rdsum1;
      o.rdsum[2] = rdsum2;
      o.rdsum[3] = 0.0;
    }
  }
  if (wave == 0 && lane < 3) {
    o.ddpose[0][lane] = dd0;                           // entry (0, l)
    o.ddpose[1][lane] = dd1;                           // entry (1, l)
  }
  if constexpr (CHAIN) {
    if (rec_wave && lane < 16) o.posefin[lane >> 2][lane & 3] = ((lane >> 2) < 3 && (lane & 3) < 3) ? Pc[lane >> 2][lane & 3] : 0.0;
  }
}

// What the solve leaves to the panel launch (its workgroup 0 of every trajectory, lanes 0..2 of wave 0): the new ranks'
// entries at the pose's state indices, zero ranks up to the bank's rank count `nrp` (a multiple of 4: the busiest
// trajectory's ranks, padded to a whole k-tile), the pose block's share of the in-place prediction, the active bound the
// next covariance pass reads, and (trajectory 0) the work-queue heads of the row-slab pass, which start every pass at zero.
// nrp == 0 -- no trajectory of the bank observed anything in this cadence, no pass will follow for it -- : the predictions'
// noise goes to the pose diagonal here (what k_predict_rc does for a single prediction-only step).
// (chained runs, small launches) the panel launch is its own gate: every workgroup waits for the cadence's solve to have
// completed (announced by the chain launch behind that solve), then drops what its L2 may hold of the records' previous use.  Only where every workgroup of the launch has a CU to itself and the solve workgroups theirs (ekf_api.hip): a waiting
// workgroup must not keep the solve it waits for from being placed; larger launches get the one-lane gate launch (k_gate).
__device__ __forceinline__ void panel_head_wait(unsigned* sync, unsigned sigma, unsigned* flags) {
  if (threadIdx.x == 0) {
    if (!sync_wait(sync + SYNC_SOLVE * SYNC_STRIDE, sigma)) atomicOr(flags + blockIdx.y, EKF_FLAG_INTERNAL);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// `tail_word` (chained runs): the covariance pass behind this launch rewrites P_base, which the gather workgroups of the next
// chain launch -- on the other stream -- may still be reading: the launch does not end before they have counted themselves
// off (one lane of the whole launch waits; bounded).
__device__ __forceinline__ void pose_epilogue(const CadOut& o, double* Pb, double* Vb, double* Wb, SolveOut* so, unsigned* queue,
                                              int b, int ld, int lane, int nrp, const unsigned* tail_word = nullptr,
                                              unsigned tail_target = 0u, unsigned* flags = nullptr, unsigned start_sigma = 0u) {
  if (tail_word && b == 0 && lane == 0) {
    bool ok = sync_wait(tail_word, tail_target);
    // ... nor before the next cadence's solve has been placed (tail_word - SYNC_GATHER + SYNC_START: the same counter block)
    if (start_sigma) ok = sync_wait(tail_word + (SYNC_START - SYNC_GATHER) * SYNC_STRIDE, start_sigma) && ok;
    if (!ok) atomicOr(flags, EKF_FLAG_INTERNAL);
  }
  const int ld16 = ld >> 4;
  const int s0 = CAD_SLOTS - o.nslots;
  // one (slot, pose index) pair per lane and round: the loads of a round are in flight together
  const int pairs = 3 * o.nslots;
  for (int e = lane; e < pairs; e += 64) {
    const int q = e / 3, l = e - 3 * q;
    const double4_t vw = *reinterpret_cast<const double4_t*>(o.posevw[s0 + q][l]);
    Vb[(long)(2 * q) * ld + l] = vw[0];
    Vb[(long)(2 * q + 1) * ld + l] = vw[1];
    Wb[wm_index(ld16, 2 * q, l)] = vw[2];
    Wb[wm_index(ld16, 2 * q + 1, l)] = vw[3];
  }
  if (lane < 3) {
    for (int k = 2 * o.nslots; k < nrp; ++k) {         // this trajectory used fewer ranks than the bank's busiest: zeros
      Vb[(long)k * ld + lane] = 0.0;
      Wb[wm_index(ld16, k, lane)] = 0.0;
    }
    double d0 = o.ddpose[0][lane], d1 = o.ddpose[1][lane];
    if (nrp == 0) {                                    // (uniform) no pass follows: the noise of the predictions, now
      if (lane == 0) d0 += o.rdsum[0];
      if (lane == 1) d1 += o.rdsum[1];
      if (lane == 2) Pb[2 * p_lds(ld) + 2] += o.rdsum[2];
    }
    Pb[lane] += d0;                                    // entry (0, l)
    if (lane >= 1) Pb[p_lds(ld) + lane] += d1;         // entry (1, l); (1, 0) lies below the diagonal
  }
  if (lane == 0) so[b].neff = o.neff;                  // what the covariance pass reads as this trajectory's bound
  if (b == 0 && lane < 8) queue[lane * RS_QSTRIDE] = 0u;
}

// ---------------------------------------------------------------------------------------------
// k_gather_cad (look-ahead): the block P[C_u, C_u] of the NEXT cadence while the ranks of this one are still pending --
//     P(a, b) = P_base[a][b] + sum_k W[a][k] V[k][b] + [a == b < 3] dacc[a]      (a <= b;  P(b, a) := P(a, b))
// -- written to gbuf, so that that cadence's solve no longer depends on the covariance pass in between and can run
// beside it (small launches: the pass leaves CUs free).  The factors at C_u are 83 x 80 scattered entries of W and as
// many of V, and a CU takes about a cycle per cache line it touches: one workgroup per trajectory spent 22 us on them.
// So the ranks are dealt over CAD_GP workgroups per trajectory, 8 ranks (two MFMA k-tiles) each: a workgroup stages
// Wc[a][k] = W[C_u[a]][k], Vc[k][a] = V[k][C_u[a]] for its ranks, forms its share of M = Wc Vc on the matrix cores,
// and writes G_p[r][l] = M[r][l] where P(C_u[r], C_u[l]) is stored that way round, M[l][r] where it is stored mirrored
// (part 0 adds the base entries and the pending pose noise); the solve adds the parts in a fixed order.  The positions
// C_u are formed exactly as k_solve_cad forms them.
// ---------------------------------------------------------------------------------------------
constexpr int CAD_GP = KTOT / 8;        // parts (workgroups per trajectory): 8 ranks each
constexpr int CAD_GW = 8;               // waves of a gather workgroup
constexpr int CAD_VS = 96 + 1;          // row stride of Vc and of M

__global__ __launch_bounds__(64 * CAD_GW) void k_gather_cad(const double* __restrict__ P, const double* __restrict__ V,
                                                            const double* __restrict__ W, const double* __restrict__ dacc,
                                                            const StepIn* __restrict__ in, const CadPlan* __restrict__ plan,
                                                            int batch, int kb,
                                                            DeviceConfig cfg, int ld, long pstride,
                                                            double* __restrict__ gbuf) {
  using G = CadGeom;
  constexpr int CU = G::CU;
  __shared__ __attribute__((aligned(16))) double Wc[96][9];          // [a][k], 8 ranks (stride 9: rows on different banks)
  __shared__ __attribute__((aligned(16))) double Vc[8][CAD_VS];      // [k][a]
  __shared__ __attribute__((aligned(16))) double Ms[96][CAD_VS];     // this part's share of M
  __shared__ int Cs[128];
  __shared__ int cntS[CAD_SLOTS + 1], firstS[CAD_SLOTS + 1], loS[CAD_SLOTS + 1];
  const int part = blockIdx.x, b = blockIdx.y;
  const int k0 = 8 * part;                             // this workgroup's ranks: k0 .. k0 + 7
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double* Pb = P + (long)b * pstride;
  const double* Vb = V + (long)b * KTOT * ld;
  const double* Wb = W + (long)b * KTOT * ld;
  const int ld16 = ld >> 4;
  const CadPlan pl = plan[b];
  cad_positions<false>(pl, in, batch, b, cfg, tid, Cs, cntS, firstS, loS, [](int, double, double) {});   // (as in k_solve_cad)
  __syncthreads();
  const int Cl0 = Cs[lane], Cl1 = Cs[64 + lane];
  // (part 0) the base entries first: their latency hides under the staging and the product
  constexpr int RQ = (CU + CAD_GW - 1) / CAD_GW;       // rows per wave
  double gv0[RQ], gv1[RQ];
#pragma unroll
  for (int q = 0; q < RQ; ++q) {
    const int r = wave + CAD_GW * q;
    gv0[q] = 0.0;
    gv1[q] = 0.0;
    if (part == 0 && r < CU) {                         // (wave-uniform)
      const int Cr = Cs[r];
      gv0[q] = Pb[p_index(ld, min(Cr, Cl0), max(Cr, Cl0))];
      if (CU > 64) gv1[q] = Pb[p_index(ld, min(Cr, Cl1), max(Cr, Cl1))];
    }
  }
  // the factors at C_u for this part's ranks: lane = (position within a group of 8, rank); both loads of both groups of a
  // thread are in flight before the first LDS store
  {
    const int kk = lane & 7, k = k0 + kk;
    const int kc = min(k, max(kb - 1, 0));             // (ranks beyond the pending ones: the last one again, masked below)
    double wv[2], vv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int a = 8 * (wave + CAD_GW * u) + (lane >> 3);   // (a < 128)
      const int row = a < CU ? Cs[a] : 0;
      wv[u] = Wb[wm_index(ld16, kc, row)];
      vv[u] = Vb[(long)kc * ld + row];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int a = 8 * (wave + CAD_GW * u) + (lane >> 3);
      if (a < 96) {
        const bool in_k = a < CU && k < kb;
        Wc[a][kk] = in_k ? wv[u] : 0.0;
        Vc[kk][a] = in_k ? vv[u] : 0.0;
      }
    }
  }
  __syncthreads();
  // M = Wc Vc: 6 x 6 tiles of 16 x 16 dealt to the 8 waves, two k-tiles each: A[i = li][k = lq] = Wc[16 rt + li][4 kt + lq],
  // B[k = lq][j = li] = Vc[4 kt + lq][16 ct + li]
  const int li = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int t = wave + CAD_GW * u;
    if (t < 36) {                                      // (wave-uniform)
      const int rt = t / 6, ct = t - 6 * rt;
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wc[16 * rt + li][lq], Vc[lq][16 * ct + li], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wc[16 * rt + li][4 + lq], Vc[4 + lq][16 * ct + li], acc, 0, 0, 0);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) Ms[16 * rt + lq + 4 * reg][16 * ct + li] = acc[reg];
    }
  }
  __syncthreads();
  const double d0 = part == 0 ? dacc[4 * b] : 0.0, d1 = part == 0 ? dacc[4 * b + 1] : 0.0, d2 = part == 0 ? dacc[4 * b + 2] : 0.0;
  double* gb = gbuf + ((long)part * batch + b) * CAD_ROWS * CAD_CS;
#pragma unroll
  for (int q = 0; q < RQ; ++q) {
    const int r = wave + CAD_GW * q;
    if (r < CU) {
      const int Cr = Cs[r];
      {
        const int l = min(lane, CU - 1);
        double v = gv0[q] + ((Cr <= Cl0) ? Ms[r][l] : Ms[l][r]);
        if (Cr == Cl0 && Cr < 3) v += Cr == 0 ? d0 : (Cr == 1 ? d1 : d2);
        gb[(long)r * CAD_CS + lane] = v;
      }
      if (64 + lane < CAD_CS) {
        const int l = min(64 + lane, CU - 1);
        double v = gv1[q] + ((Cr <= Cl1) ? Ms[r][l] : Ms[l][r]);
        if (Cr == Cl1 && Cr < 3) v += Cr == 0 ? d0 : (Cr == 1 ? d1 : d2);
        gb[(long)r * CAD_CS + 64 + lane] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_panels_cad: thread i (a state index >= 3) replays the whole cadence on its column X[a] = P(C_u[a], i).
// NW waves of 64 state indices per workgroup share one staging of the records (LDS, 16-byte broadcast reads at
// compile-time offsets); after the single barrier the waves never synchronise again.
// ---------------------------------------------------------------------------------------------
template <class F, int... S>
__device__ __forceinline__ void static_for_slots(F&& f, std::integer_sequence<int, S...>) {
  (f(std::integral_constant<int, S>{}), ...);
}
// x -= k[lane 16 r + L of every row of 16 lanes r] * e: the fp64 DPP broadcast of this part (v_fmac_f64_dpp row_newbcast, the
// negation a source modifier): the same fused operation as fma(-k, e, x), with k read from ANOTHER lane's register
template <int L>
__device__ __forceinline__ void fnmac_row_bcast(double& x, double k, double e) {
  asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(k), "v"(e), "n"(L));
}

// Round 6: the K rows of a landmark's down-date are not read as one LDS broadcast each (83 ds_read_b128 for the first landmark,
// 1 720 per wave and cadence: the compute phase was bound by what the LDS returns, 64 lanes x 16 B per read) but SIXTEEN ROWS
// PER READ -- lane l takes row 16 g + (l & 15) of group g -- and handed to the FMAs by the DPP broadcast within each row of 16
// lanes: <= 6 reads per landmark; N = 2000 x 32: 75 - 77 -> 66 - 68 us (profiles/r06_panel_launch.txt).  The same fused
// operations in the same order as the broadcast reads of k_panels_cad_ks: the shapes still agree bit for bit.
#ifdef CADP_STAMPS                                      /* diagnostic build: s_memtime stamps of one wave of the panel launch, printed */
#define PSTAMP(k)                                                                  \
  do {                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pst_[k])::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                             \
  } while (0)
#else
#define PSTAMP(k) do { } while (0)
#endif
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_panels_cad(double* __restrict__ P, double* __restrict__ V,
                                                        double* __restrict__ W, const double* __restrict__ mu_in,
                                                        double* __restrict__ mu_out, const int* __restrict__ nact,
                                                        const CadOut* __restrict__ co, SolveOut* __restrict__ so,
                                                        unsigned* __restrict__ queue, int ld, long pstride, int nrp,
                                                        const double* __restrict__ colbuf, double* __restrict__ prow3,
                                                        unsigned* __restrict__ sync, unsigned head_sigma, unsigned tail_target,
                                                        unsigned* __restrict__ flags, unsigned start_sigma, int skipw) {
  // (`skipw`, "w_from_v": the covariance pass behind this launch forms its W fragments from V and the records' S^-1 -- W is
  //  half of what this launch writes, and it is bound by what it writes; the pose's entries, pose_epilogue, are kept)
  using G = CadGeom;
  constexpr int CU = G::CU, GM = G::GM, NT = 64 * NW;
  const unsigned* tail_word = sync ? sync + SYNC_GATHER * SYNC_STRIDE : nullptr;
  if (sync && head_sigma) panel_head_wait(sync, head_sigma, flags);
  __shared__ __attribute__((aligned(16))) double sRec[G::REC + 32];   // (+ what the last records' grouped K reads overshoot)
  __shared__ double2 sG[CAD_SLOTS];
  __shared__ int sF[CAD_SLOTS];
  const int b = blockIdx.y;
  const int n = nact[b];
  const int w0 = blockIdx.x * NT;
  if (w0 >= n) return;
  const CadOut& o = co[b];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nslots = o.nslots, npred = o.npred, neff = o.neff;
  const int s0 = GM - nslots;                          // slots in use: s0 .. GM - 1
  const int ld16 = ld >> 4;
  double* Pb = P + (long)b * pstride;
  double* Vb = V + (long)b * KTOT * ld;
  double* Wb = W + (long)b * KTOT * ld;
  const int i0 = w0 + wave * 64, i = i0 + lane;
  const bool act = i < n;
  const int ii = act ? i : n - 1;                      // idle lanes shadow the last state index (no stores)
  const bool actw = act && i >= 3;                     // the pose's state indices are the solve's
  const bool busy = nslots > 0 || npred > 0;           // (uniform) this trajectory does something in this cadence
  const bool live = i0 < neff && i0 < n && busy;       // (uniform) this wave replays
#ifdef CADP_STAMPS
  unsigned long long pst_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  PSTAMP(0);
#ifdef CADP_STAGGER                                     /* diagnostic build: every other workgroup starts late (x 3.4 us) */
  if (NW == 4 && (blockIdx.x & 1)) {
#pragma unroll
    for (int z = 0; z < CADP_STAGGER; ++z) __builtin_amdgcn_s_sleep(127);
  }
#endif
  // The records of the cadence (what a workgroup shares) are REQUESTED first, then the gather -- nothing of it depends on the
  // staged records -- and only then are the records written to LDS: one memory round trip for both.  (Round 6.  The staging loop
  // used to follow the gather and wait for every load in flight at each of its eight iterations: nine round trips.)
  const bool stage = w0 < neff && busy;                // (uniform) some wave of this workgroup replays
  constexpr int SQ = 8;                                // staged 16-byte entries per thread and round
  const double2* rsrc = reinterpret_cast<const double2*>(o.rec);
  double2* rdst = reinterpret_cast<double2*>(sRec);
  const int e0s = G::rec_off(s0) / 2 + tid;
  double stx[SQ], sty[SQ];                             // (plain doubles: arrays of double2 end up in scratch)
  int sf = 0;
  double sgx = 0.0, sgy = 0.0;
  if (stage) {
#pragma unroll
    for (int q = 0; q < SQ; ++q) {
      const double2 t = rsrc[min(e0s + q * NT, G::REC / 2 - 1)];
      stx[q] = t.x;
      sty[q] = t.y;
    }
    if (tid < CAD_SLOTS) {
      sf = o.sfirst[tid];
      const double2 t = *reinterpret_cast<const double2*>(o.g[tid]);
      sgx = t.x;
      sgy = t.y;
    }
  }
  // The gather: ONE load per entry at a selected address, every one of them independent of the others -- 83 loads in flight.
  // (A branch per pair of positions -- by whether the pair is mirrored for the whole wave, lies in the row direction or
  //  straddles it -- merged its loads in one temporary, and every pair waited for the pair before it: 40 round trips.)
  // A landmark's two positions are two ADJACENT state indices c, c + 1.  For the state indices i <= c the entries P(c, i),
  // P(c + 1, i) are stored mirrored, as P_base(i, c), P_base(i, c + 1): a different cache line per lane (a CU takes about a
  // cycle per line).  Where the column gather beside the solve has laid a pair down as rows (colbuf: pairs that are mirrored for
  // the whole 64-index strip and do not straddle two column panels) it is read from there, coalesced and past the caches.
  double X[CU];
  if (live) {
    const long row_ii = (long)ii * p_lds(ld), col_ii = p_col(ld, ii);
    auto gather = [&](auto with_colbuf) {
      constexpr bool CB = decltype(with_colbuf)::value;
#pragma unroll
      for (int a = 0; a < 3; ++a) X[a] = __builtin_nontemporal_load(Pb + (ii >= a ? (long)a * p_lds(ld) + col_ii : row_ii + a));
#pragma unroll
      for (int a = 3; a < CU; a += 2) {
        const int c0 = o.C[a], c1 = o.C[a + 1];
#ifdef CADP_SKIP_GATHER                                 /* diagnostic build: every gather reads the row direction */
        X[a] = Pb[p_index(ld, min(c0, 2), ii)];
        X[a + 1] = Pb[p_index(ld, min(c1, 2), ii)];
#else
        // entry (c, ii) for ii >= c, (ii, c) below it (positions beyond the cadence's carry index 0: row 0, which nothing uses)
        const long u0 = (long)c0 * p_lds(ld) + col_ii, l0 = row_ii + p_col(ld, c0);
        const long u1 = (long)c1 * p_lds(ld) + col_ii, l1 = row_ii + p_col(ld, c1);
        const double* q0 = Pb + (ii >= c0 ? u0 : l0);
        const double* q1 = Pb + (ii >= c1 ? u1 : l1);
        if constexpr (CB) {
          if (c1 == c0 + 1 && i0 + 63 <= c0 && (c1 & (PPW - 1)) != 0) {   // (uniform: an address, not a load, is chosen)
            q0 = colbuf + ((long)b * CAD_CU + a) * ld + ii;
            q1 = q0 + ld;
          }
          X[a] = __builtin_nontemporal_load(q0);       // (each entry is read exactly once)
          X[a + 1] = __builtin_nontemporal_load(q1);
        } else {
          X[a] = *q0;                                  // (without the column gather the mirrored pairs of neighbouring landmarks
          X[a + 1] = *q1;                              //  share cache lines: through the caches)
        }
#endif
      }
    };
    if (colbuf) gather(std::true_type{});
    else gather(std::false_type{});
  } else {
#pragma unroll
    for (int a = 0; a < CU; ++a) X[a] = 0.0;
  }
  PSTAMP(1);
  if (stage) {
#pragma unroll
    for (int q = 0; q < SQ; ++q)
      if (e0s + q * NT < G::REC / 2) rdst[e0s + q * NT] = make_double2(stx[q], sty[q]);
    for (int eb = e0s + SQ * NT; eb < G::REC / 2; eb += SQ * NT) {   // (one-wave workgroups: the rest, SQ requests at a time)
      double tx[SQ], ty[SQ];
#pragma unroll
      for (int q = 0; q < SQ; ++q) {
        const double2 t = rsrc[min(eb + q * NT, G::REC / 2 - 1)];
        tx[q] = t.x;
        ty[q] = t.y;
      }
#pragma unroll
      for (int q = 0; q < SQ; ++q)
        if (eb + q * NT < G::REC / 2) rdst[eb + q * NT] = make_double2(tx[q], ty[q]);
    }
    if (tid < 16) rdst[G::REC / 2 + tid] = make_double2(0.0, 0.0);
    if (tid < CAD_SLOTS) {
      sF[tid] = sf;
      sG[tid] = make_double2(sgx, sgy);
    }
  }
  PSTAMP(2);
  __syncthreads();
  PSTAMP(3);
  if (i0 >= n) return;
  if (!live) {
    // beyond the active bound the rows and columns of P are exactly zero off the diagonal (and an idle trajectory appends
    // nothing): the cadence's ranks are zero there and the mean is carried over
    if (actw) {
      for (int k = 0; k < nrp; ++k) {
        Vb[(long)k * ld + i] = 0.0;
        if (!skipw) Wb[wm_index(ld16, k, i)] = 0.0;
      }
      mu_out[(long)b * ld + i] = mu_in[(long)b * ld + i];
      if (prow3) {                                     // (chained runs) the pose rows as they stand: nothing of this wave changes
#pragma unroll
        for (int a = 0; a < 3; ++a) prow3[((long)b * 3 + a) * ld + i] = Pb[p_index(ld, a, i)];
      }
    }
    if (blockIdx.x == 0 && wave == 0) pose_epilogue(o, Pb, Vb, Wb, so, queue, b, ld, lane, nrp, tail_word, tail_target, flags, start_sigma);
    return;
  }
  double d0 = 0.0, d1 = 0.0, dm = 0.0;
  int tp = 0;                                          // (uniform) next touched step whose prediction is due
  // prediction (:430): of the stored triangle it changes rows 0, 1 only; thread i adds its two entries to P_base at the end
  // (the ranks are unaffected), the pose diagonal's noise never meets a state index >= 3
  auto predictions_before = [&](int s) {
    while (tp < npred && sF[tp] <= s) {
      const double2 g = sG[tp];
      const double t0 = g.x * X[2], t1 = g.y * X[2];
      X[0] += t0;
      X[1] += t1;
      d0 += t0;
      d1 += t1;
      ++tp;
    }
  };
  // (the slots as a compile-time sequence: s, and with it every index of X and of the records, is a constant)
  auto slot = [&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if constexpr (s == 0) PSTAMP(4);
    if constexpr (s == 1) PSTAMP(5);
    if constexpr (s == 10) PSTAMP(6);
    if constexpr (s == 20) PSTAMP(7);
    if constexpr (s == 30) PSTAMP(8);
    if (s >= s0) {                                     // (uniform)
      predictions_before(s);
      constexpr int pa = G::pa(s), off = G::rec_off(s);
      const int kr = 2 * (s - s0);
      const double2* R = reinterpret_cast<const double2*>(__builtin_assume_aligned(sRec + off, 16));
      constexpr int NG = (pa + 15) / 16;
      double2 kg[NG];
      {                                                // the K rows, sixteen per read: issued in front of the e chain
        const double2* Rl = R + 8 + (lane & 15);
#pragma unroll
        for (int g = 0; g < NG; ++g) kg[g] = Rl[16 * g];
      }
      double2 hk[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) hk[k] = R[k];
      const double2 s01 = R[5], s23 = R[6], yy = R[7];
      double e0 = hk[0].x * X[0], e1 = hk[0].y * X[0];      // (H_s P_s)[:, i] = h5 . x[sel]
#pragma unroll
      for (int k = 1; k < 5; ++k) {
        const double xv = (k < 3) ? X[k] : X[pa + (k - 3)];
        e0 = fma(hk[k].x, xv, e0);
        e1 = fma(hk[k].y, xv, e1);
      }
      const double f0 = e0 * s01.x + e1 * s23.x;       // K_s[i, :] = (H_s P_s)[:, i]^T S^-1  (P symmetric)
      const double f1 = e0 * s01.y + e1 * s23.y;
      dm += f0 * yy.x + f1 * yy.y;                     // :476
#ifdef CADP_SKIP_STORE                                  /* diagnostic build: no rank stores (a value that is never -7 keeps e, f alive) */
      if (actw && e0 == -7.0 && f0 == -7.0) {
#else
      if (actw) {
#endif
        Vb[(long)kr * ld + i] = e0;
        Vb[(long)(kr + 1) * ld + i] = e1;
        if (!skipw) {                                  // (uniform)
          Wb[wm_index(ld16, kr, i)] = -f0;
          Wb[wm_index(ld16, kr + 1, i)] = -f1;
        }
      }
      // x[a] -= K_s[C_u[a], :] . (H_s P_s)[:, i], what lives on (behind the last slot: the pose rows, for the predictions of
      // steps that observe nothing)
#ifdef CADP_SKIP_DD                                     /* diagnostic build: only the rows the next landmark reads are down-dated */
      constexpr int ND = pa < 5 ? pa : 5;
#else
      constexpr int ND = pa;
#endif
      // X[a] = fma(-K[a].x, e0, X[a]); X[a] = fma(-K[a].y, e1, X[a]) -- per group of 16 rows first every e0 term, then every e1
      // term (the order within a row is what it was; two dependent DPP operations back to back cost a wait state each)
      static_for_slots([&](auto gc) {
        constexpr int g = decltype(gc)::value, NA = (ND - 16 * g) < 16 ? (ND - 16 * g) : 16;
        static_for_slots([&](auto ac) {
          constexpr int a = 16 * g + decltype(ac)::value;
          fnmac_row_bcast<(a & 15)>(X[a], kg[g].x, e0);
        }, std::make_integer_sequence<int, NA>{});
        static_for_slots([&](auto ac) {
          constexpr int a = 16 * g + decltype(ac)::value;
          fnmac_row_bcast<(a & 15)>(X[a], kg[g].y, e1);
        }, std::make_integer_sequence<int, NA>{});
      }, std::make_integer_sequence<int, (ND + 15) / 16>{});
    }
  };
  static_for_slots(slot, std::make_integer_sequence<int, GM>{});
  PSTAMP(9);
  predictions_before(GM);                              // steps behind the last landmark
  if (actw) {
    for (int k = 2 * nslots; k < nrp; ++k) {           // fewer ranks than the bank's busiest trajectory (and the k-tile pad): zeros
      Vb[(long)k * ld + i] = 0.0;
      if (!skipw) Wb[wm_index(ld16, k, i)] = 0.0;
    }
    Pb[p_col(ld, i)] += d0;                            // entry (0, i)
    Pb[p_col(ld, i) + p_lds(ld)] += d1;                // entry (1, i)
    mu_out[(long)b * ld + i] = mu_in[(long)b * ld + i] + dm;
    if (prow3) {                                       // (chained runs) P(0..2, i) after the cadence: what the replay ends with
#pragma unroll
      for (int a = 0; a < 3; ++a) prow3[((long)b * 3 + a) * ld + i] = X[a];
    }
  }
#ifdef CADP_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PSTAMP(10);
  if (NW == 4 && (blockIdx.x == 7 || blockIdx.x == 1) && (b == 5 || b == 20) && tid == 64 && nslots == GM)
    printf("PST bx %d b %d: gather-issued %llu staged %llu barrier %llu slot0 %llu slot1 %llu slot10 %llu slot20 %llu slot30 %llu end-slots %llu drained %llu\n",
           (int)blockIdx.x, b, pst_[1] - pst_[0], pst_[2] - pst_[0], pst_[3] - pst_[0], pst_[4] - pst_[0], pst_[5] - pst_[0], pst_[6] - pst_[0],
           pst_[7] - pst_[0], pst_[8] - pst_[0], pst_[9] - pst_[0], pst_[10] - pst_[0]);
#endif
  if (blockIdx.x == 0 && wave == 0) pose_epilogue(o, Pb, Vb, Wb, so, queue, b, ld, lane, nrp, tail_word, tail_target, flags, start_sigma);
}

// ---------------------------------------------------------------------------------------------
// k_panels_cad_ks: the latency form of the panel launch (few state indices: one trajectory at N = 2000 is 63 waves on
// 256 CUs, and one wave's replay is a serial chain of 40 landmarks x up to 83 rows).  The four waves of a workgroup
// share 64 state indices and split the ROWS of X: every wave keeps the pose rows, landmark position-slot q (positions
// 3 + 2q, 4 + 2q) belongs to wave q & 3.  Per landmark the owner of its two rows finishes e = (H P)[:, i] -- the same
// chain of five FMAs as k_panels_cad, so the two forms agree bit for bit -- and hands it to the others through LDS
// (double-buffered by landmark parity: one barrier per landmark); each wave then down-dates its own rows (<= 23 instead
// of 83).  Rows of a wave sit at position 3 + 8p + 2 wave + e: one base address per wave, compile-time offsets.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_panels_cad_ks(double* __restrict__ P, double* __restrict__ V,
                                                       double* __restrict__ W, const double* __restrict__ mu_in,
                                                       double* __restrict__ mu_out, const int* __restrict__ nact,
                                                       const CadOut* __restrict__ co, SolveOut* __restrict__ so,
                                                       unsigned* __restrict__ queue, int ld, long pstride, int nrp,
                                                       const double* __restrict__ colbuf, double* __restrict__ prow3,
                                                       unsigned* __restrict__ sync, unsigned head_sigma, unsigned tail_target,
                                                       unsigned* __restrict__ flags, unsigned start_sigma) {
  using G = CadGeom;
  constexpr int GM = G::GM, CU = G::CU;
  constexpr int LP = (GM + 3) / 4;                     // landmark position-slots per wave
  const unsigned* tail_word = sync ? sync + SYNC_GATHER * SYNC_STRIDE : nullptr;
  if (sync && head_sigma) panel_head_wait(sync, head_sigma, flags);
  __shared__ __attribute__((aligned(16))) double sRec[G::REC + 32];   // (+ what the last record's K reads may overshoot)
  __shared__ double2 sG[CAD_SLOTS];
  __shared__ int sF[CAD_SLOTS];
  __shared__ double2 sE[2][64];
  const int b = blockIdx.y;
  const int n = nact[b];
  const int i0 = blockIdx.x * 64;
  if (i0 >= n) return;
  const CadOut& o = co[b];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nslots = o.nslots, npred = o.npred, neff = o.neff;
  const int s0 = GM - nslots;                          // slots in use: s0 .. GM - 1
  const int ld16 = ld >> 4;
  double* Pb = P + (long)b * pstride;
  double* Vb = V + (long)b * KTOT * ld;
  double* Wb = W + (long)b * KTOT * ld;
  const int i = i0 + lane;
  const bool act = i < n;
  const int ii = act ? i : n - 1;                      // idle lanes shadow the last state index (no stores)
  const bool actw = act && i >= 3;                     // the pose's state indices are the solve's
  const bool live = i0 < neff && (nslots > 0 || npred > 0);   // (uniform) this workgroup replays
  // (as in k_panels_cad: the records are requested first, then the gather -- one independent load per entry at a selected
  //  address --, and only then are the records written to LDS: one memory round trip for everything)
  double XP[3], XL[2 * LP];
  if (live) {
    constexpr int SQ = 8;
    const double2* src = reinterpret_cast<const double2*>(o.rec);
    double2* dst = reinterpret_cast<double2*>(sRec);
    const int e0s = G::rec_off(s0) / 2 + tid;
    double stx[SQ], sty[SQ];
#pragma unroll
    for (int q = 0; q < SQ; ++q) {
      const double2 t = src[min(e0s + q * 256, G::REC / 2 - 1)];
      stx[q] = t.x;
      sty[q] = t.y;
    }
    int sf = 0;
    double sgx = 0.0, sgy = 0.0;
    if (tid < CAD_SLOTS) {
      sf = o.sfirst[tid];
      const double2 t = *reinterpret_cast<const double2*>(o.g[tid]);
      sgx = t.x;
      sgy = t.y;
    }
    const long row_ii = (long)ii * p_lds(ld), col_ii = p_col(ld, ii);
#pragma unroll
    for (int k = 0; k < 3; ++k) XP[k] = Pb[ii >= k ? (long)k * p_lds(ld) + col_ii : row_ii + k];
#pragma unroll
    for (int pp = 0; pp < LP; ++pp) {
      const int a = 3 + 8 * pp + 2 * wave;
      const int c0 = o.C[min(a, CU)], c1 = o.C[min(a + 1, CU)];   // (C[CU] = 0: a position beyond the cadence's)
      const double* q0 = Pb + (ii >= c0 ? (long)c0 * p_lds(ld) + col_ii : row_ii + p_col(ld, c0));
      const double* q1 = Pb + (ii >= c1 ? (long)c1 * p_lds(ld) + col_ii : row_ii + p_col(ld, c1));
      if (colbuf && a < CU && c1 == c0 + 1 && i0 + 63 <= c0 && (c1 & (PPW - 1)) != 0) {   // (uniform) gathered beside the solve
        q0 = colbuf + ((long)b * CAD_CU + a) * ld + ii;
        q1 = q0 + ld;
      }
      XL[2 * pp] = *q0;
      XL[2 * pp + 1] = *q1;
    }
#pragma unroll
    for (int q = 0; q < SQ; ++q)
      if (e0s + q * 256 < G::REC / 2) dst[e0s + q * 256] = make_double2(stx[q], sty[q]);
    if (tid < 16) dst[G::REC / 2 + tid] = make_double2(0.0, 0.0);
    if (tid < CAD_SLOTS) {
      sF[tid] = sf;
      sG[tid] = make_double2(sgx, sgy);
    }
  }
  __syncthreads();
  if (!live) {
    // beyond the active bound the rows and columns of P are exactly zero off the diagonal (and an idle trajectory appends
    // nothing): the cadence's ranks are zero there and the mean is carried over
    if (actw && wave == 0) {
      for (int k = 0; k < nrp; ++k) {
        Vb[(long)k * ld + i] = 0.0;
        Wb[wm_index(ld16, k, i)] = 0.0;
      }
      mu_out[(long)b * ld + i] = mu_in[(long)b * ld + i];
      if (prow3) {                                     // (chained runs) the pose rows as they stand
#pragma unroll
        for (int a = 0; a < 3; ++a) prow3[((long)b * 3 + a) * ld + i] = Pb[p_index(ld, a, i)];
      }
    }
    if (blockIdx.x == 0 && wave == 0) pose_epilogue(o, Pb, Vb, Wb, so, queue, b, ld, lane, nrp, tail_word, tail_target, flags, start_sigma);
    return;
  }
  const double* recw = sRec + 4 * wave;                // K of this wave's rows: + compile-time offsets
  double d0 = 0.0, d1 = 0.0, dm = 0.0;
  int par = 0;                                         // (uniform) hand-over buffer: alternates with every landmark processed
  int tp = 0;                                          // (uniform) next touched step whose prediction is due
  auto predictions_before = [&](int s) {               // prediction (:430): rows 0, 1 (see k_panels_cad)
    while (tp < npred && sF[tp] <= s) {
      const double2 g = sG[tp];
      const double t0 = g.x * XP[2], t1 = g.y * XP[2];
      XP[0] += t0;
      XP[1] += t1;
      d0 += t0;
      d1 += t1;
      ++tp;
    }
  };
#pragma unroll
  for (int s = 0; s < GM; ++s) {                       // (compile-time after unrolling)
    if (s >= s0) {                                     // (uniform)
      predictions_before(s);
      const int pa = G::pa(s), off = G::rec_off(s);
      const int kr = 2 * (s - s0);
      const int q = GM - 1 - s, OW = q & 3, PL = q >> 2;   // the landmark's position-slot: owner wave, its local pair
      const double2* R = reinterpret_cast<const double2*>(__builtin_assume_aligned(sRec + off, 16));
      if (wave == OW) {                                // (uniform) e = (H_s P_s)[:, i] = h5 . x[sel], finished by the rows' owner
        double2 hk[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) hk[k] = R[k];
        double e0 = hk[0].x * XP[0], e1 = hk[0].y * XP[0];
#pragma unroll
        for (int k = 1; k < 5; ++k) {
          const double xv = (k < 3) ? XP[k] : XL[2 * PL + (k - 3)];
          e0 = fma(hk[k].x, xv, e0);
          e1 = fma(hk[k].y, xv, e1);
        }
        sE[par][lane] = make_double2(e0, e1);
      }
      WG_LDS_BARRIER();
      const double2 ee = sE[par][lane];
      par ^= 1;
      const double e0 = ee.x, e1 = ee.y;
      const double2 s01 = R[5], s23 = R[6], yy = R[7];
      const double f0 = e0 * s01.x + e1 * s23.x;       // K_s[i, :] = (H_s P_s)[:, i]^T S^-1  (P symmetric)
      const double f1 = e0 * s01.y + e1 * s23.y;
      dm += f0 * yy.x + f1 * yy.y;                     // :476
      if (actw && wave == (s & 3)) {                   // (the ranks' stores dealt over the waves)
        Vb[(long)kr * ld + i] = e0;
        Vb[(long)(kr + 1) * ld + i] = e1;
        Wb[wm_index(ld16, kr, i)] = -f0;
        Wb[wm_index(ld16, kr + 1, i)] = -f1;
      }
      // x[a] -= K_s[C_u[a], :] . (H_s P_s)[:, i], what lives on (behind the last slot: the pose rows only)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double2 kc = R[8 + k];
        XP[k] = fma(-kc.x, e0, XP[k]);
        XP[k] = fma(-kc.y, e1, XP[k]);
      }
      // own landmark rows at positions 3 + 8p + 2 wave + e < pa; the pair at the boundary may already be dead for
      // this wave: what lands in a dead row does not matter
      const double2* Rw = reinterpret_cast<const double2*>(__builtin_assume_aligned(recw + off + 16, 16));
#pragma unroll
      for (int pp = 0; pp < LP; ++pp) {
        if (3 + 8 * pp < pa) {                         // (compile-time) some wave's rows of this pair are live
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const double2 kc = Rw[3 + 8 * pp + e];
            XL[2 * pp + e] = fma(-kc.x, e0, XL[2 * pp + e]);
            XL[2 * pp + e] = fma(-kc.y, e1, XL[2 * pp + e]);
          }
        }
      }
    }
  }
  predictions_before(GM);                              // steps behind the last landmark
  if (actw && wave == 0) {
    for (int k = 2 * nslots; k < nrp; ++k) {           // fewer ranks than the bank's busiest trajectory (and the k-tile pad): zeros
      Vb[(long)k * ld + i] = 0.0;
      Wb[wm_index(ld16, k, i)] = 0.0;
    }
    Pb[p_col(ld, i)] += d0;                            // entry (0, i)
    Pb[p_col(ld, i) + p_lds(ld)] += d1;                // entry (1, i)
    mu_out[(long)b * ld + i] = mu_in[(long)b * ld + i] + dm;
    if (prow3) {                                       // (chained runs) P(0..2, i) after the cadence
#pragma unroll
      for (int a = 0; a < 3; ++a) prow3[((long)b * 3 + a) * ld + i] = XP[a];
    }
  }
  if (blockIdx.x == 0 && wave == 0) pose_epilogue(o, Pb, Vb, Wb, so, queue, b, ld, lane, nrp, tail_word, tail_target, flags, start_sigma);
}

// ---------------------------------------------------------------------------------------------
// k_chain_cad (round 6, chained solves): the block P[C', C'] and the mean at C' of the NEXT cadence from the RECORDS of this
// one (per landmark H, S^-1, y, K at the positions; per step the motion Jacobian; the pose block behind the last landmark) and
// the covariance as it stood BEFORE this cadence (P_base behind the previous pass; its rows 0..2 from `prow3`, which the
// previous panel launch left -- this cadence's own panel launch changes rows 0, 1 of P_base in place and may be running).
// The only true dependency between two cadences of a trajectory is the sequential landmark recurrence
// (src/replay_no_ros.py:436-480); this kernel is what lets the solves follow one another on the handle's stream while panel
// launch and covariance pass of every cadence run on the second one.
//
// A cadence is a LINEAR map on the rows P_0(C_u, .) it starts from (C_u: its positions): with M_q the product of its
// predictions and (I - K H) factors in front of landmark q, restricted to C_u (83 x 83, M_0 = I), landmark q's two rank rows are
//     V[2q .. 2q+1][i] = T_q P_0(C_u, i),   T_q = H_q M_q   (2 x 83)            for every state index i >= 3,
// and rows 0..2 of M behind the cadence give the pose rows.  M telescopes (M_{q+1} = M_q - K_q T_q; a prediction adds g x row 2
// to rows 0, 1), so the T_q satisfy a unit lower block-triangular system whose coefficients are all in the records:
//     T_q = A_q - sum_{r<q} C_{q,r} T_r,
//     C_{q,r} = H_q[:, 0..2] G^{(q,r)} K_r[0..2, :] + H_q[:, 3..4] K_r[pa_q .. pa_q + 1, :]      (2 x 2)
//     A_q     = H_q[:, 0..2] G^{(q,-1)} at the pose positions, H_q[:, 3..4] at positions pa_q, pa_q + 1
// (G^{(q,r)}: the predictions between landmark r and landmark q -- their g add up, they only ever add multiples of row 2).
// The rank rows at the next positions C' are wanted, E = T X with X = P_0(C_u, C'): the kernel solves for them DIRECTLY,
//     (I + C) E = A X          (A X: five rows of X per landmark -- the product T X never forms)
// -- the 780 coefficient blocks in parallel, the inverses of the five 16 x 16 diagonal blocks by substitution on the identity
// (Linv_i; Linv_i C_{i,<i} and Linv_i (A X)_i as products), then five block steps on the fp64 matrix cores, a column tile of
// the 80 landmark positions of C' per wave -- and gets the pose rows behind the cadence as three more rows of the same
// product (rows 0..2 of M).  Then
//     F  = -S^-1 E  (per landmark, 2 x 2)    = W at C'
//     P(C', C') = P_0(C', C') + F^T E        landmark x landmark (the upper block triangle: 15 tiles);  pose rows / columns from
//                                            E's last three rows, the pose block from the solve (its upper triangle);
//     mean(C') = mean_0(C') - sum_q F_q^T y_q
// X and P_0(C', C') -- 13.8 k scattered entries -- are fetched by GATHER workgroups of the same launch while the chain
// workgroup forms its coefficients; a POSITIONS workgroup forms the inputs of the cadence after the next (CadPre).
// Operands in LDS.  One chain workgroup per trajectory.
// ---------------------------------------------------------------------------------------------
// Diagnostic build (-DCHAIN_STAMPS): s_memtime stamps of k_chain_cad's phases (wave 0 of trajectory 0) behind the means in gmu
// (batch x 128 doubles, then 32 stamps); tools/chain_stamps.py reads them through ekf_debug_snapshot(which = 5).
#ifdef CHAIN_STAMPS
#define CHSTAMP(k)                                                                               \
  do {                                                                                           \
    if (wave == 0 && b == 0) {                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      unsigned long long t_;                                                                     \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      if (lane == 0) reinterpret_cast<unsigned long long*>(gmu + (long)batch * 128)[k] = t_;     \
    }                                                                                            \
  } while (0)
#else
#define CHSTAMP(k) do { } while (0)
#endif
constexpr int CH_NC = 80;               // columns: the landmark positions of the next cadence (5 MFMA tiles)
constexpr int CH_S = 81;                // LDS row stride of A (odd: rows and columns both spread over the banks)
constexpr int CH_R = 96;                // rows of the padded operands (6 MFMA tiles: 80 rank rows, the 3 pose rows)
constexpr int CH_CS = 82;               // LDS row stride of B / the coefficient matrix (even: a 2 x 2 block's row is one 16-byte read)
constexpr int CH_GW = 12;               // gather workgroups per trajectory at most (12: a row of X and of P_0(C', C') per wave)

// ---- pieces shared by k_chain_cad and k_panels_cad_tf: the cadence's records as the triangular system (I + C) E = A X ----
struct ChainRec {                       // (LDS) the small parts of the records, by landmark q = slot s0k + q
  double2 hS[CAD_SLOTS][5];             // H_q: {H[0][k], H[1][k]}
  double siS[CAD_SLOTS][4];             // S_q^-1, row-major
  double2 yS[CAD_SLOTS];                // innovation
  double2 pgS[CAD_SLOTS + 1];           // predictions in front of landmark q, summed from the cadence's start; [nk]: all
};
// threads 0 .. 39 and 64 .. 104 of the workgroup; a barrier behind it
__device__ __forceinline__ void chain_stage_records(const CadOut& op, int nk, int s0k, ChainRec& R, int tid) {
  using G = CadGeom;
  double2 (&hS)[CAD_SLOTS][5] = R.hS;
  double (&siS)[CAD_SLOTS][4] = R.siS;
  double2 (&yS)[CAD_SLOTS] = R.yS;
  double2 (&pgS)[CAD_SLOTS + 1] = R.pgS;
  if (tid < CAD_SLOTS) {
    double4_t si = {0.0, 0.0, 0.0, 0.0};
    double2 y = make_double2(0.0, 0.0);
    double2 hh[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) hh[k] = make_double2(0.0, 0.0);
    if (tid < nk) {                                    // landmark tid = slot s0k + tid
      const double* rec = op.rec + G::rec_off(s0k + tid);
#pragma unroll
      for (int k = 0; k < 5; ++k) hh[k] = *reinterpret_cast<const double2*>(rec + 2 * k);
      const double2 sa = *reinterpret_cast<const double2*>(rec + 10), sb = *reinterpret_cast<const double2*>(rec + 12);
      si = double4_t{sa.x, sa.y, sb.x, sb.y};
      y = *reinterpret_cast<const double2*>(rec + 14);
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) hS[tid][k] = hh[k];
    siS[tid][0] = si[0];
    siS[tid][1] = si[1];
    siS[tid][2] = si[2];
    siS[tid][3] = si[3];
    yS[tid] = y;
  }
  if (tid >= 64 && tid <= 64 + CAD_SLOTS) {
    // the panel launch applies touched step t's prediction in front of slot sfirst[t] (k_panels_cad: predictions_before);
    // predictions only ever add multiples of row 2 to rows 0, 1, so their g add up
    const int q = tid - 64, np = min(op.npred, CAD_SLOTS);
    double g0 = 0.0, g1 = 0.0;
    for (int t = 0; t < np; ++t) {
      if (q == nk || op.sfirst[t] <= s0k + q) {
        g0 += op.g[t][0];
        g1 += op.g[t][1];
      }
    }
    pgS[q] = make_double2(g0, g1);
  }
}
// the 2 x 2 blocks C_{q,r}, r < q, and the rows behind the cadence (q = nk) into Cm (zeroed, a barrier in front); all threads
__device__ __forceinline__ void chain_coefficient_blocks(const CadOut& op, int nk, int s0k, const ChainRec& R, double (*Cm)[CH_CS],
                                                         int tid) {
  using G = CadGeom;
  const double2 (&hS)[CAD_SLOTS][5] = R.hS;
  const double2 (&pgS)[CAD_SLOTS + 1] = R.pgS;
  {
    const int tri = nk * (nk - 1) / 2, items = tri + nk;
    for (int e = tid; e < items; e += 64 * CAD_NW) {
      int q, r;
      if (e < tri) {
        q = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)e)) * 0.5f);   // q (q - 1) / 2 <= e < q (q + 1) / 2
        while (q * (q - 1) / 2 > e) --q;
        while (q * (q + 1) / 2 <= e) ++q;
        r = e - q * (q - 1) / 2;
      } else {
        q = nk;
        r = e - tri;
      }
      const double* rr = op.rec + G::rec_off(s0k + r) + 16;          // K_r[a] = (rr[2a], rr[2a + 1])
      const double2 k0 = *reinterpret_cast<const double2*>(rr), k1 = *reinterpret_cast<const double2*>(rr + 2),
                    k2 = *reinterpret_cast<const double2*>(rr + 4);
      const double2 gq = pgS[q], gr = pgS[r];
      const double g0 = gq.x - gr.x, g1 = gq.y - gr.y;              // the predictions between landmark r and landmark q
      if (q < nk) {
        const int paq = G::pa(s0k + q);
        const double2 h0 = hS[q][0], h1 = hS[q][1], h2 = hS[q][2], h3 = hS[q][3], h4 = hS[q][4];
        const double2 ka = *reinterpret_cast<const double2*>(rr + 2 * paq), kb = *reinterpret_cast<const double2*>(rr + 2 * paq + 2);
        // row w of H_q[:, 0..2] G = (h[w][0], h[w][1], h[w][2] + g0 h[w][0] + g1 h[w][1])
        const double hx2 = fma(g0, h0.x, fma(g1, h1.x, h2.x)), hy2 = fma(g0, h0.y, fma(g1, h1.y, h2.y));
        Cm[2 * q][2 * r] = h0.x * k0.x + h1.x * k1.x + hx2 * k2.x + h3.x * ka.x + h4.x * kb.x;
        Cm[2 * q][2 * r + 1] = h0.x * k0.y + h1.x * k1.y + hx2 * k2.y + h3.x * ka.y + h4.x * kb.y;
        Cm[2 * q + 1][2 * r] = h0.y * k0.x + h1.y * k1.x + hy2 * k2.x + h3.y * ka.x + h4.y * kb.x;
        Cm[2 * q + 1][2 * r + 1] = h0.y * k0.y + h1.y * k1.y + hy2 * k2.y + h3.y * ka.y + h4.y * kb.y;
      } else {                                                      // rows 0..2 of G^{(end, r)} K_r[0..2, :]
        Cm[KTOT][2 * r] = fma(g0, k2.x, k0.x);
        Cm[KTOT][2 * r + 1] = fma(g0, k2.y, k0.y);
        Cm[KTOT + 1][2 * r] = fma(g1, k2.x, k1.x);
        Cm[KTOT + 1][2 * r + 1] = fma(g1, k2.y, k1.y);
        Cm[KTOT + 2][2 * r] = k2.x;
        Cm[KTOT + 2][2 * r + 1] = k2.y;
        // rows 83, 84 (the panel form of the launch): what the cadence's predictions alone add to rows 0, 1 -- the in-place
        // share of P_base(0, i), P_base(1, i) beside the ranks -- is  ge_p X[2]  minus these coefficients times E
        Cm[KTOT + 3][2 * r] = g0 * k2.x;
        Cm[KTOT + 3][2 * r + 1] = g0 * k2.y;
        Cm[KTOT + 4][2 * r] = g1 * k2.x;
        Cm[KTOT + 4][2 * r + 1] = g1 * k2.y;
      }
    }
  }
}
// Linv_i = (I + C_ii)^-1 by substitution on the identity (wave i < 5, a column per lane); barriers around it are the caller's
__device__ __forceinline__ void chain_invert_diagonal(const double (*Cm)[CH_CS], double (*Li)[16][17], int wave, int lane) {
  if (wave < 5 && lane < 16) {
    const int i = wave, c = lane;
    double t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = k == c ? 1.0 : 0.0;
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      double2 c0[7], c1[7];
#pragma unroll
      for (int jp = 0; jp < j; ++jp) {                 // the landmark's coefficients within the block: all reads in flight together
        c0[jp] = *reinterpret_cast<const double2*>(&Cm[16 * i + 2 * j][16 * i + 2 * jp]);
        c1[jp] = *reinterpret_cast<const double2*>(&Cm[16 * i + 2 * j + 1][16 * i + 2 * jp]);
      }
#pragma unroll
      for (int jp = 0; jp < j; ++jp) {
        t[2 * j] = fma(-c0[jp].x, t[2 * jp], t[2 * j]);
        t[2 * j + 1] = fma(-c1[jp].x, t[2 * jp], t[2 * j + 1]);
        t[2 * j] = fma(-c0[jp].y, t[2 * jp + 1], t[2 * j]);
        t[2 * j + 1] = fma(-c1[jp].y, t[2 * jp + 1], t[2 * j + 1]);
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) Li[i][k][c] = t[k];
  }
}
// Linv_i C_{i,<i} in place (tiles of 16 x 16 over the waves): a block step of the solve becomes ONE product
__device__ __forceinline__ void chain_scale_blocks(double (*Cm)[CH_CS], const double (*Li)[16][17], int nk, int wave, int lane) {
  const int li = lane & 15, lq = lane >> 4;
  for (int job = wave; job < 10; job += CAD_NW) {      // (i, j), j < i <= 4: tile (rows of block i, columns of block j)
    int i = 1, j = job;
    while (j >= i) {
      j -= i;
      ++i;
    }
    if (8 * i < nk) {                                  // (uniform)
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[i][li][4 * kt + lq], Cm[16 * i + 4 * kt + lq][16 * j + li], acc, 0, 0, 0);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) Cm[16 * i + lq + 4 * reg][16 * j + li] = acc[reg];
    }
  }
}

__global__ __launch_bounds__(64 * CAD_NW) void k_chain_cad(const double* __restrict__ P, const double* __restrict__ prow3,
                                                           const double* __restrict__ mu_land, const double* __restrict__ mu_pose,
                                                           const CadOut* __restrict__ prev, const StepIn* __restrict__ in,
                                                           const CadPlan* __restrict__ plan, int batch, DeviceConfig cfg, int ld,
                                                           long pstride, double* __restrict__ gbuf, double* __restrict__ gmu,
                                                           double* __restrict__ xg, double* __restrict__ bg,
                                                           unsigned* __restrict__ sync, unsigned gather_target,
                                                           unsigned* __restrict__ flags, int gw, unsigned sigma,
                                                           const CadPre* __restrict__ pre_in, CadPre* __restrict__ pre_out,
                                                           const CadPlan* __restrict__ plan2, int wait_pass) {
  using G = CadGeom;
  constexpr int CU = G::CU;
  __shared__ int Cs[128], Ck[128];
  __shared__ int cntS[CAD_SLOTS + 1], firstS[CAD_SLOTS + 1], loS[CAD_SLOTS + 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)blockIdx.x >= batch * (1 + gw)) {
    // ---- the POSITIONS workgroups of the launch (one per trajectory, where another cadence follows the next one): the inputs of
    // the cadence AFTER the next as its solve, its chain launch and that launch's gather workgroups need them (CadPre) ----
    const int b = (int)blockIdx.x - batch * (1 + gw);
    __shared__ double2 zP[CAD_SLOTS];
    const CadPlan pl2 = plan2[b];
    if (tid < CAD_SLOTS) zP[tid] = make_double2(0.0, 0.0);
    __syncthreads();
    const int ns2 = cad_positions<true>(pl2, in, batch, b, cfg, tid, Cs, cntS, firstS, loS,
                                        [&](int s, double zr, double zb) { zP[s] = make_double2(zr, zb); });
    __syncthreads();
    CadPre& po = pre_out[b];
    if (tid < 128) po.C[tid] = Cs[tid];
    if (tid <= CAD_SLOTS) {
      po.cnt[tid] = cntS[tid];
      po.first[tid] = firstS[tid];
      po.lo[tid] = loS[tid];
    }
    if (tid < CAD_SLOTS) {
      int fl = 0;
      double lin = 0.0, ang = 0.0;
      if (tid < pl2.ns) {
        const StepIn& st = in[(long)(pl2.t0 + tid) * batch + b];
        fl = st.flags;
        if (tid == 0 && pl2.j0 > 0) fl &= ~FLAG_PREDICT;   // a step cut by the previous cadence: its prediction has happened
        lin = st.lin;
        ang = st.ang;
      }
      po.fl[tid] = fl;
      po.la[tid][0] = lin;
      po.la[tid][1] = ang;
      po.z[tid][0] = zP[tid].x;
      po.z[tid][1] = zP[tid].y;
    }
    if (tid == 0) po.nslots = ns2;
    return;
  }
  if ((int)blockIdx.x >= batch) {
    // ---- the GATHER workgroups of the launch: X = P_0(C_u, C') and P_0(C', C') as coalesced rows (xg, bg: batch x 84 x 88) ----
    // 13.8 k scattered 8-byte entries per trajectory, and a CU takes about a cycle per cache line it touches: 13 us on the one CU
    // of the chain workgroup.  Nothing of it depends on the records' arithmetic: CH_GW workgroups per trajectory fetch 7 rows
    // of each array apiece (`gw` = CH_GW where the chip has room) while the chain workgroup forms its coefficients, write them through and count themselves off.
    const int g = (int)blockIdx.x - batch, b = g / gw, part = g - b * gw;
    const double* Pb = P + (long)b * pstride;
    const CadOut& op = prev[b];
    const int cuk = 3 + 2 * min(op.nslots, CAD_SLOTS);
    if (tid < 128) Ck[tid] = tid < CU ? op.C[tid] : 0;
    int ns1;
    if (pre_in) {                                      // (uniform) formed one cadence ahead
      if (tid < 128) Cs[tid] = pre_in[b].C[tid];
      ns1 = pre_in[b].nslots;
    } else {
      ns1 = cad_positions<false>(plan[b], in, batch, b, cfg, tid, Cs, cntS, firstS, loS, [](int, double, double) {});
    }
    const int cu = 3 + 2 * ns1;
    // P_base and the pose rows are what the previous covariance pass (and the panel launch in front of it) left: the mark launch
    // behind that pass has said so (sigma - 1; `wait_pass` = 0: no chained pass is in flight, the stream's order covers it); the
    // entries are read past this XCD's L2, which may hold older ones
    if (tid == 0 && wait_pass && !sync_wait(sync + SYNC_PASS * SYNC_STRIDE, sigma - 1u)) atomicOr(flags + b, EKF_FLAG_INTERNAL);
    __syncthreads();
    for (int r = part + gw * wave; r < CAD_ROWS; r += gw * CAD_NW) {   // (uniform) rows part, part + gw, ... dealt over the waves
      double x[2] = {0.0, 0.0}, pb[2] = {0.0, 0.0};
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int l = 64 * hf + lane;
        if (l >= 3 && l < cu) {
          const int Cl = Cs[l];
          if (r < cuk) {
            const int Cr = Ck[r];
            x[hf] = r < 3 ? ld_dev(prow3 + ((long)b * 3 + r) * ld + Cl) : ld_dev(Pb + p_index(ld, min(Cr, Cl), max(Cr, Cl)));
          }
          if (r >= 3 && r < cu) {
            const int Cr = Cs[r];
            pb[hf] = ld_dev(Pb + p_index(ld, min(Cr, Cl), max(Cr, Cl)));
          }
        }
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int l = 64 * hf + lane;
        if (l < CAD_CS) {
          st_dev(xg + ((long)b * CAD_ROWS + r) * CAD_CS + l, x[hf]);
          st_dev(bg + ((long)b * CAD_ROWS + r) * CAD_CS + l, pb[hf]);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // written through ...
    __syncthreads();                                   // ... by every wave ...
    if (tid == 0) __hip_atomic_fetch_add(sync + SYNC_GATHER * SYNC_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... then counted
    return;
  }
  // Columns: the 80 LANDMARK positions of C' (column j = position 3 + j; the pose columns of X are never used): 5 MFMA tiles.
  __shared__ __attribute__((aligned(16))) double A[CH_R][CH_S];      // A X -> E (rows 0..79), the pose rows behind the cadence (80..82)
  __shared__ __attribute__((aligned(16))) double B[CH_R][CH_CS];     // the coefficients C -> Linv C  ->  -F  ->  F^T E
  __shared__ __attribute__((aligned(16))) double Li[5][16][17];      // inverses of the diagonal blocks I + C_ii
  __shared__ ChainRec R;
  double2 (&hS)[CAD_SLOTS][5] = R.hS;
  double (&siS)[CAD_SLOTS][4] = R.siS;
  double2 (&yS)[CAD_SLOTS] = R.yS;
  double2 (&pgS)[CAD_SLOTS + 1] = R.pgS;
  __shared__ double dmS[5][CH_NC];                     // partial sums of the mean update, 8 landmarks each
  double (*Cm)[CH_CS] = B;
  const int b = blockIdx.x;
  const CadOut& op = prev[b];
  const CadPlan pl = plan[b];
  const int nk = min(op.nslots, CAD_SLOTS), s0k = CAD_SLOTS - nk;   // this cadence: landmarks, first slot
  // this launch runs: the solve in front of it on the stream has completed -- what the gate on the other stream waits for
  if (tid == 0) __hip_atomic_store(sync + SYNC_SOLVE * SYNC_STRIDE, sigma, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  CHSTAMP(0);
  chain_stage_records(op, nk, s0k, R, tid);
  int nslots;
  if (pre_in) {                                        // (uniform) the next cadence's positions, formed one cadence ahead
    if (tid < 128) Cs[tid] = pre_in[b].C[tid];
    nslots = pre_in[b].nslots;
  } else {
    nslots = cad_positions<false>(pl, in, batch, b, cfg, tid, Cs, cntS, firstS, loS, [](int, double, double) {});
  }
  const int cu = 3 + 2 * nslots;                       // the next cadence: positions in use
  // ---- the coefficients of (I + C) E = A X: zero, then the 2 x 2 blocks (q, r), r < q, and the pose rows (q = nk) ----
  {
    double2* z = reinterpret_cast<double2*>(&Cm[0][0]);
    for (int e = tid; e < CH_R * CH_CS / 2; e += 64 * CAD_NW) z[e] = make_double2(0.0, 0.0);
  }
  __syncthreads();
  CHSTAMP(1);
  chain_coefficient_blocks(op, nk, s0k, R, Cm, tid);
  __syncthreads();
  CHSTAMP(2);
  const int li = lane & 15, lq = lane >> 4;
  // ---- the diagonal blocks: Linv_i = (I + C_ii)^-1 by substitution on the identity (wave i, a column per lane), then
  // Linv_i C_{i,<i} in place (tiles of 16 x 16 over the waves) -- so that a block step of the solve is ONE product ----
  if (tid == 64 * (CAD_NW - 1)) {
    // (meanwhile, one lane of an idle wave) the gathered rows are there: the launch's gather workgroups have counted themselves off
    if (!sync_wait(sync + SYNC_GATHER * SYNC_STRIDE, gather_target)) atomicOr(flags + b, EKF_FLAG_INTERNAL);
  }
  chain_invert_diagonal(Cm, Li, wave, lane);
  __syncthreads();
  CHSTAMP(3);
  // ---- the gathered rows: everything this workgroup reads of them is requested here, the products below run under the
  // round trip ----
  constexpr int RQ = (CAD_ROWS + CAD_NW - 1) / CAD_NW;               // rows per wave (11)
  const double* bgb = bg + (long)b * CAD_ROWS * CAD_CS;
  double p0[RQ][2];                                                  // P_0(C', C') in the layout the block is written in
#pragma unroll
  for (int q = 0; q < RQ; ++q) {
    const int r = wave + CAD_NW * q;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int l = 64 * hf + lane;
      p0[q][hf] = (r >= 3 && r < cu && l >= 3 && l < cu) ? ld_dev(bgb + (long)r * CAD_CS + l) : 0.0;
    }
  }
  const double* xgb = xg + (long)b * CAD_ROWS * CAD_CS + 3;          // column j = position 3 + j
  const int ja = lane, jb = 64 + lane;                 // columns; the second part: 16 lanes
  const bool hb = jb < CH_NC;
  double xa[3], xb[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    xa[k] = ld_dev(xgb + (long)k * CAD_CS + ja);
    xb[k] = hb ? ld_dev(xgb + (long)k * CAD_CS + jb) : 0.0;
  }
  constexpr int QW = CAD_SLOTS / CAD_NW;               // landmarks per wave (5)
  double la0[QW], la1[QW], lb0[QW], lb1[QW];
#pragma unroll
  for (int u = 0; u < QW; ++u) {
    const int q = wave + CAD_NW * u;
    const int paq = q < nk ? G::pa(s0k + q) : 3;       // (uniform)
    la0[u] = ld_dev(xgb + (long)paq * CAD_CS + ja);
    la1[u] = ld_dev(xgb + (long)(paq + 1) * CAD_CS + ja);
    lb0[u] = hb ? ld_dev(xgb + (long)paq * CAD_CS + jb) : 0.0;
    lb1[u] = hb ? ld_dev(xgb + (long)(paq + 1) * CAD_CS + jb) : 0.0;
  }
  chain_scale_blocks(Cm, Li, nk, wave, lane);
  CHSTAMP(4);
  // ---- the right-hand side A X, row by row: A_q = H_q[:, 0..2] G^{(q,-1)} at the pose positions, H_q[:, 3..4] at the landmark's
  // own -- five rows of X per landmark, each landmark row of X read exactly once (straight from xg, coalesced); the pose rows
  // behind the cadence start from G^{(end,-1)} X[0..2].  Columns over lanes (64 + 16), landmarks over waves ----
  {
#pragma unroll
    for (int u = 0; u < QW; ++u) {
      const int q = wave + CAD_NW * u;
      double2 h[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) h[k] = hS[q][k];     // (zeros beyond the cadence's landmarks)
      const double2 gq = pgS[min(q, nk)];
      const double hx2 = fma(gq.x, h[0].x, fma(gq.y, h[1].x, h[2].x)), hy2 = fma(gq.x, h[0].y, fma(gq.y, h[1].y, h[2].y));
      A[2 * q][ja] = h[0].x * xa[0] + h[1].x * xa[1] + hx2 * xa[2] + h[3].x * la0[u] + h[4].x * la1[u];
      A[2 * q + 1][ja] = h[0].y * xa[0] + h[1].y * xa[1] + hy2 * xa[2] + h[3].y * la0[u] + h[4].y * la1[u];
      if (hb) {
        A[2 * q][jb] = h[0].x * xb[0] + h[1].x * xb[1] + hx2 * xb[2] + h[3].x * lb0[u] + h[4].x * lb1[u];
        A[2 * q + 1][jb] = h[0].y * xb[0] + h[1].y * xb[1] + hy2 * xb[2] + h[3].y * lb0[u] + h[4].y * lb1[u];
      }
    }
    if (wave < 2) {                                    // rows 80 .. 95: the pose rows behind the cadence, then zeros
      const double2 ge = pgS[nk];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = KTOT + 8 * wave + k;
        const double va = r == KTOT ? fma(ge.x, xa[2], xa[0]) : (r == KTOT + 1 ? fma(ge.y, xa[2], xa[1]) : (r == KTOT + 2 ? xa[2] : 0.0));
        const double vb = r == KTOT ? fma(ge.x, xb[2], xb[0]) : (r == KTOT + 1 ? fma(ge.y, xb[2], xb[1]) : (r == KTOT + 2 ? xb[2] : 0.0));
        A[r][ja] = va;
        if (hb) A[r][jb] = vb;
      }
    }
  }
  __syncthreads();
  CHSTAMP(5);
  // ---- Linv_i (A X)_i in place: 5 blocks x 5 column tiles over the waves ----
  for (int job = wave; job < 25; job += CAD_NW) {
    const int i = job / 5, ct = job - 5 * i;
    if (8 * i < nk) {                                  // (uniform)
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[i][li][4 * kt + lq], A[16 * i + 4 * kt + lq][16 * ct + li], acc, 0, 0, 0);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) A[16 * i + lq + 4 * reg][16 * ct + li] = acc[reg];
    }
  }
  __syncthreads();
  CHSTAMP(6);
  // ---- the block steps: E_i = Linv_i (A X)_i - (Linv_i C_{i,<i}) E_{<i}; i = 5: the pose rows A_end X - C_end E.  One column
  // tile per wave (waves 0..4), the landmarks solved so far as k ----
  for (int i = 1; i <= 5; ++i) {
    if (i == 5 || 8 * i < nk) {                        // (uniform) something of this block is in use
      const int kts = i == 5 ? (2 * nk + 3) / 4 : 4 * i;
      if (wave < 5) {
        double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        int kt = 0;
        for (; kt + 1 < kts; kt += 2) {
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Cm[16 * i + li][4 * kt + lq], A[4 * kt + lq][16 * wave + li], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Cm[16 * i + li][4 * kt + 4 + lq], A[4 * kt + 4 + lq][16 * wave + li], acc1, 0, 0, 0);
        }
        if (kt < kts)
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Cm[16 * i + li][4 * kt + lq], A[4 * kt + lq][16 * wave + li], acc0, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) A[16 * i + lq + 4 * reg][16 * wave + li] -= acc0[reg] + acc1[reg];
      }
      __syncthreads();
    }
  }
  CHSTAMP(7);
  // ---- -F = -(E S^-1) per landmark: f = e S^-1 is K at the position (P symmetric), W = -f; and the mean update's partial sums ----
  for (int e = tid; e < 5 * CH_NC; e += 64 * CAD_NW) {
    const int g8 = e / CH_NC, c = e - g8 * CH_NC;
    double dm = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int q = 8 * g8 + k;
      const double e0 = A[2 * q][c], e1 = A[2 * q + 1][c];
      const double f0 = e0 * siS[q][0] + e1 * siS[q][2], f1 = e0 * siS[q][1] + e1 * siS[q][3];
      B[2 * q][c] = -f0;
      B[2 * q + 1][c] = -f1;
      const double2 y = yS[q];
      dm = fma(f0, y.x, dm);
      dm = fma(f1, y.y, dm);
    }
    dmS[g8][c] = dm;
  }
  __syncthreads();
  CHSTAMP(8);
  // the mean at C' (src/replay_no_ros.py:476 summed over the cadence): mean_0 + sum_q f_q . y_q
  if (tid < 128) {
    double v = 0.0;
    if (tid < 3) v = mu_pose[(long)b * ld + tid];
    else if (tid < cu) {
      const int j = tid - 3;
      v = ld_dev(mu_land + (long)b * ld + Cs[tid]) + ((((dmS[0][j] + dmS[1][j]) + dmS[2][j]) + dmS[3][j]) + dmS[4][j]);
    }
    gmu[(long)b * 128 + tid] = v;
  }
  // ---- F^T E, the upper block triangle: 15 tiles of 16 x 16 over the 8 waves, 20 k-tiles; A-operand (-F)^T ----
  constexpr int TPW = 2;                               // tiles per wave at most
  double4_t acc[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    acc[u] = double4_t{0.0, 0.0, 0.0, 0.0};
    const int t = wave + CAD_NW * u;
    if (t < 15) {                                      // (uniform) tile t of the triangle, row tile rt <= column tile ct
      int rt = 0, left = t;
      while (left >= 5 - rt) {
        left -= 5 - rt;
        ++rt;
      }
      const int ct = rt + left;
#pragma unroll 5
      for (int kt = 0; kt < KTOT / 4; ++kt)
        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(B[4 * kt + lq][16 * rt + li], A[4 * kt + lq][16 * ct + li], acc[u], 0, 0, 0);
    }
  }
  __syncthreads();
  CHSTAMP(9);
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    const int t = wave + CAD_NW * u;
    if (t < 15) {
      int rt = 0, left = t;
      while (left >= 5 - rt) {
        left -= 5 - rt;
        ++rt;
      }
      const int ct = rt + left;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) B[16 * rt + lq + 4 * reg][16 * ct + li] = acc[u][reg];
    }
  }
  __syncthreads();
  // ---- the block, in the layout k_solve_cad reads its look-ahead parts in (one part): landmark x landmark entries from the
  // product's upper triangle BY POSITION (exactly symmetric), pose rows / columns from E's last three rows, the pose block from
  // the solve ----
  double* gb = gbuf + (long)b * CAD_ROWS * CAD_CS;
#pragma unroll
  for (int q = 0; q < RQ; ++q) {
    const int r = wave + CAD_NW * q;
    if (r < CAD_ROWS) {                                // (uniform)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int l = 64 * hf + lane;
        if (l < CAD_CS) {
          double v = 0.0;
          if (r < cu && l < cu) {
            // (the pose block's UPPER triangle, like every gather from P_base: the solve's simple-form arithmetic keeps its block
            //  symmetric to rounding only, and an antisymmetric part handed from solve to solve grows -- x 1.16 per cadence at
            //  N = 40, 1e-16 -> 1e-4 in 1200 steps; profiles/r06_chained_solves.txt)
            if (r < 3 && l < 3) v = op.posefin[min(r, l)][max(r, l)];
            else if (r < 3) v = A[KTOT + r][l - 3];
            else if (l < 3) v = A[KTOT + l][r - 3];
            else v = p0[q][hf] + B[min(r, l) - 3][max(r, l) - 3];
          }
          gb[(long)r * CAD_CS + l] = v;
        }
      }
    }
  }
  CHSTAMP(10);
}

// ---------------------------------------------------------------------------------------------
// k_panels_cad_tf (round 6): the panel launch of a CHAINED cadence in the latency regime, as the same triangular solve
// k_chain_cad uses -- the cadence's rank rows at the state indices i of this workgroup are  E = (I + C)^-1 (A X)  with
// X = P_0(C_u, i)  -- instead of the replay of its 40 landmarks one after the other (k_panels_cad_ks: 0.7 us per landmark,
// 27 us for a single trajectory's panel launch whatever the number of CUs it has to itself).  One 8-wave workgroup per 64 state
// indices: every workgroup forms the coefficients C, the inverses of their diagonal blocks and Linv C from the records itself
// (as k_chain_cad does beside it: nothing to hand over), gathers its 83 x 64 entries of X meanwhile, forms A X (five rows of X
// per landmark) in LDS and takes five block steps on the matrix cores, 4 column tiles on 4 waves -- ~70 dependent MFMAs.
// Rows of the product: 0..79 the rank rows (V; W = -S^-1-scaled), 80..82 the pose rows behind the cadence (-> prow3 for the
// next chained block), 83..84 what the cadence's predictions add to P_base(0, i), P_base(1, i) in place.  Same algebra as
// the replay in a different order of summation: equal to rounding (tests/test_gpu_cadence.py), not bit for bit.
// ---------------------------------------------------------------------------------------------
constexpr int TF_S = 65;                // LDS row stride of the right-hand sides (64 state indices)

__global__ __launch_bounds__(64 * CAD_NW) void k_panels_cad_tf(double* __restrict__ P, double* __restrict__ V,
                                                               double* __restrict__ W, const double* __restrict__ mu_in,
                                                               double* __restrict__ mu_out, const int* __restrict__ nact,
                                                               const CadOut* __restrict__ co, SolveOut* __restrict__ so,
                                                               unsigned* __restrict__ queue, int ld, long pstride, int nrp,
                                                               double* __restrict__ prow3, unsigned* __restrict__ sync,
                                                               unsigned head_sigma, unsigned tail_target,
                                                               unsigned* __restrict__ flags, unsigned start_sigma) {
  using G = CadGeom;
  constexpr int GM = G::GM, CU = G::CU;
  __shared__ __attribute__((aligned(16))) double A[CH_R][TF_S];      // A X -> the rows of the product
  __shared__ __attribute__((aligned(16))) double Cm[CH_R][CH_CS];    // the coefficients C -> Linv C
  __shared__ __attribute__((aligned(16))) double Li[5][16][17];
  __shared__ ChainRec R;
  __shared__ double xpS[3][64];                        // X[0..2][i]: the pose rows of this workgroup's state indices
  __shared__ double dmS[5][64];
  __shared__ int Ck[128];
  const unsigned* tail_word = sync ? sync + SYNC_GATHER * SYNC_STRIDE : nullptr;
  if (sync && head_sigma) panel_head_wait(sync, head_sigma, flags);
  const int b = blockIdx.y;
  const int n = nact[b];
  const int i0 = blockIdx.x * 64;
  if (i0 >= n) return;
  const CadOut& o = co[b];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nslots = min(o.nslots, GM), npred = o.npred, neff = o.neff;
  const int s0 = GM - nslots;
  const int ld16 = ld >> 4;
  double* Pb = P + (long)b * pstride;
  double* Vb = V + (long)b * KTOT * ld;
  double* Wb = W + (long)b * KTOT * ld;
  const int i = i0 + lane;
  const bool act = i < n;
  const int ii = act ? i : n - 1;                      // idle lanes shadow the last state index (no stores)
  const bool actw = act && i >= 3;                     // the pose's state indices are the solve's
  const bool live = i0 < neff && (nslots > 0 || npred > 0);   // (uniform) this workgroup has something to do
  if (!live) {
    // beyond the active bound the rows and columns of P are exactly zero off the diagonal (and an idle trajectory appends
    // nothing): the cadence's ranks are zero there and the mean is carried over
    if (actw && wave == 0) {
      for (int k = 0; k < nrp; ++k) {
        Vb[(long)k * ld + i] = 0.0;
        Wb[wm_index(ld16, k, i)] = 0.0;
      }
      mu_out[(long)b * ld + i] = mu_in[(long)b * ld + i];
      if (prow3) {
#pragma unroll
        for (int a = 0; a < 3; ++a) prow3[((long)b * 3 + a) * ld + i] = Pb[p_index(ld, a, i)];
      }
    }
    if (blockIdx.x == 0 && wave == 0) pose_epilogue(o, Pb, Vb, Wb, so, queue, b, ld, lane, nrp, tail_word, tail_target, flags, start_sigma);
    return;
  }
  // ---- the gathers first (nothing of them depends on the records' arithmetic): position pair p = wave + 8 q (positions
  // 3 + 2 p, 4 + 2 p: the two state indices of ONE landmark, the rows of slot GM - 1 - p), lanes = state indices ----
  if (tid < 128) Ck[tid] = tid < CU ? o.C[tid] : 0;
  chain_stage_records(o, nslots, s0, R, tid);
  __syncthreads();
  constexpr int PQ = GM / CAD_NW;                      // pairs per wave (5)
  double x0[PQ], x1[PQ], xp = 0.0;
#pragma unroll
  for (int q = 0; q < PQ; ++q) {
    const int a = 3 + 2 * (wave + CAD_NW * q);
    const int c0 = Ck[a], c1 = Ck[a + 1];
    if (c1 == c0 + 1 && i0 + 63 <= c0 && (c1 & (PPW - 1)) != 0) {   // (uniform) both mirrored: side by side in row i (see k_panels_cad)
      const v2d_u v = *reinterpret_cast<const v2d_u*>(Pb + p_index(ld, ii, c0));
      x0[q] = v.x;
      x1[q] = v.y;
    } else {
      x0[q] = Pb[p_index(ld, min(c0, ii), max(c0, ii))];
      x1[q] = Pb[p_index(ld, min(c1, ii), max(c1, ii))];
    }
  }
  if (wave < 3) xp = Pb[p_index(ld, min(wave, ii), max(wave, ii))];
  // ---- the coefficients (k_chain_cad's, formed again here: the two launches run side by side) ----
  {
    double2* z = reinterpret_cast<double2*>(&Cm[0][0]);
    for (int e = tid; e < CH_R * CH_CS / 2; e += 64 * CAD_NW) z[e] = make_double2(0.0, 0.0);
  }
  if (wave < 3) xpS[wave][lane] = xp;
  __syncthreads();
  chain_coefficient_blocks(o, nslots, s0, R, Cm, tid);
  __syncthreads();
  chain_invert_diagonal(Cm, Li, wave, lane);
  __syncthreads();
  chain_scale_blocks(Cm, Li, nslots, wave, lane);
  // ---- A X: landmark q's rows from the pose rows and ITS pair of rows of X (the wave that gathered the pair has them) ----
  {
    const double xa0 = xpS[0][lane], xa1 = xpS[1][lane], xa2 = xpS[2][lane];
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
      const int p = wave + CAD_NW * q, sl = GM - 1 - p, lq_ = sl - s0;   // the slot at this pair, its landmark number
      if (lq_ >= 0) {                                  // (uniform) the pair is in use
        double2 h[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) h[k] = R.hS[lq_][k];
        const double2 gq = R.pgS[lq_];
        const double hx2 = fma(gq.x, h[0].x, fma(gq.y, h[1].x, h[2].x)), hy2 = fma(gq.x, h[0].y, fma(gq.y, h[1].y, h[2].y));
        A[2 * lq_][lane] = h[0].x * xa0 + h[1].x * xa1 + hx2 * xa2 + h[3].x * x0[q] + h[4].x * x1[q];
        A[2 * lq_ + 1][lane] = h[0].y * xa0 + h[1].y * xa1 + hy2 * xa2 + h[3].y * x0[q] + h[4].y * x1[q];
      } else {                                         // rows of landmarks the cadence does not have: zero
        const int zr = 2 * (nslots + (p - nslots));    // (p >= nslots: rows 2 p, 2 p + 1 are beyond the cadence's)
        A[zr][lane] = 0.0;
        A[zr + 1][lane] = 0.0;
      }
    }
    if (wave < 2) {                                    // rows 80 .. 95: the pose rows behind the cadence, the in-place rows, zeros
      const double2 ge = R.pgS[nslots];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = KTOT + 8 * wave + k;
        A[r][lane] = r == KTOT ? fma(ge.x, xa2, xa0) : (r == KTOT + 1 ? fma(ge.y, xa2, xa1) : (r == KTOT + 2 ? xa2 :
                     (r == KTOT + 3 ? ge.x * xa2 : (r == KTOT + 4 ? ge.y * xa2 : 0.0))));
      }
    }
  }
  __syncthreads();
  const int li = lane & 15, lq = lane >> 4;
  // ---- Linv_i (A X)_i in place: 5 blocks x 4 column tiles over the waves ----
  for (int job = wave; job < 20; job += CAD_NW) {
    const int bi = job >> 2, ct = job & 3;
    if (8 * bi < nslots) {                             // (uniform)
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[bi][li][4 * kt + lq], A[16 * bi + 4 * kt + lq][16 * ct + li], acc, 0, 0, 0);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) A[16 * bi + lq + 4 * reg][16 * ct + li] = acc[reg];
    }
  }
  __syncthreads();
  // ---- the block steps (a column tile per wave, waves 0..3: one per SIMD); bi = 5: the rows behind the cadence ----
  for (int bi = 1; bi <= 5; ++bi) {
    if (bi == 5 || 8 * bi < nslots) {                  // (uniform)
      const int kts = bi == 5 ? (2 * nslots + 3) / 4 : 4 * bi;
      if (wave < 4) {
        double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        int kt = 0;
        for (; kt + 1 < kts; kt += 2) {
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Cm[16 * bi + li][4 * kt + lq], A[4 * kt + lq][16 * wave + li], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Cm[16 * bi + li][4 * kt + 4 + lq], A[4 * kt + 4 + lq][16 * wave + li], acc1, 0, 0, 0);
        }
        if (kt < kts)
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Cm[16 * bi + li][4 * kt + lq], A[4 * kt + lq][16 * wave + li], acc0, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) A[16 * bi + lq + 4 * reg][16 * wave + li] -= acc0[reg] + acc1[reg];
      }
      __syncthreads();
    }
  }
  // ---- results: V rows and W (-S^-1-scaled) of the cadence's ranks, zero ranks up to the bank's count, the mean, the in-place
  // share of rows 0, 1, the pose rows behind the cadence.  Landmarks over waves (5 each), state indices over lanes ----
  {
    double dm = 0.0;
#pragma unroll
    for (int u = 0; u < PQ; ++u) {
      const int q = wave + CAD_NW * u;
      if (q < nslots) {                                // (uniform)
        const double e0 = A[2 * q][lane], e1 = A[2 * q + 1][lane];
        const double f0 = e0 * R.siS[q][0] + e1 * R.siS[q][2], f1 = e0 * R.siS[q][1] + e1 * R.siS[q][3];
        const double2 y = R.yS[q];
        dm = fma(f0, y.x, dm);
        dm = fma(f1, y.y, dm);
        if (actw) {
          Vb[(long)(2 * q) * ld + i] = e0;
          Vb[(long)(2 * q + 1) * ld + i] = e1;
          Wb[wm_index(ld16, 2 * q, i)] = -f0;
          Wb[wm_index(ld16, 2 * q + 1, i)] = -f1;
        }
      }
    }
    if (wave < 5) dmS[wave][lane] = 0.0;
    __syncthreads();
    // (the mean's sum over the landmarks in a fixed order: wave w adds into row w % 5 in two rounds)
    if (wave < 5) dmS[wave][lane] = dm;
    __syncthreads();
    if (wave >= 5) dmS[wave - 5][lane] += dm;
    __syncthreads();
  }
  if (wave == 0 && actw) {
    for (int k = 2 * nslots; k < nrp; ++k) {           // fewer ranks than the bank's busiest trajectory (and the k-tile pad): zeros
      Vb[(long)k * ld + i] = 0.0;
      Wb[wm_index(ld16, k, i)] = 0.0;
    }
    Pb[p_col(ld, i)] += A[KTOT + 3][lane];             // entry (0, i)
    Pb[p_col(ld, i) + p_lds(ld)] += A[KTOT + 4][lane]; // entry (1, i)
    mu_out[(long)b * ld + i] = mu_in[(long)b * ld + i] + ((((dmS[0][lane] + dmS[1][lane]) + dmS[2][lane]) + dmS[3][lane]) + dmS[4][lane]);
    if (prow3) {
#pragma unroll
      for (int a = 0; a < 3; ++a) prow3[((long)b * 3 + a) * ld + i] = A[KTOT + a][lane];
    }
  }
  if (blockIdx.x == 0 && wave == 0) pose_epilogue(o, Pb, Vb, Wb, so, queue, b, ld, lane, nrp, tail_word, tail_target, flags, start_sigma);
}

// (chained runs) the gate in front of a cadence's panel launch on the second stream: it ends when the cadence's solve has completed
// (announced by the chain launch behind that solve).  One lane: it cannot keep the solve from finding its CUs, whatever the
// order in which the host's enqueues reach the two streams.
__global__ void k_gate(unsigned* __restrict__ sync, unsigned sigma, unsigned* __restrict__ flags, int batch) {
  if (threadIdx.x != 0) return;
  if (!sync_wait(sync + SYNC_SOLVE * SYNC_STRIDE, sigma))
    for (int b = 0; b < batch; ++b) atomicOr(flags + b, EKF_FLAG_INTERNAL);
}

// (chained runs) behind a chained cadence's covariance pass on the second stream: "the pass of transition sigma is done" -- what
// the NEXT chain launch's gather workgroups wait for.  A launch of its own, enqueued with its pass: every wait of the chained
// order is for a launch that was enqueued EARLIER, so the order stays free of deadlock even where the runtime maps the handle's
// two streams onto ONE hardware queue (HIP multiplexes streams onto a few: then the launches simply run in the order they were
// enqueued -- no overlap, no hang).
__global__ void k_mark(unsigned* __restrict__ sync, unsigned sigma) {
  if (threadIdx.x == 0) __hip_atomic_store(sync + SYNC_PASS * SYNC_STRIDE, sigma, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (chained runs, once per run) rows 0..2 of every trajectory's P_base -> prow3: what the first chained block reads while the
// first panel launch changes rows 0, 1 in place
__global__ __launch_bounds__(256) void k_snap_pose(const double* __restrict__ P, const int* __restrict__ nact, int ld, long pstride,
                                                   double* __restrict__ prow3) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nact[b] || i < 3) return;
  const double* Pb = P + (long)b * pstride;
#pragma unroll
  for (int a = 0; a < 3; ++a) prow3[((long)b * 3 + a) * ld + i] = Pb[p_index(ld, a, i)];
}

// ---------------------------------------------------------------------------------------------
// launchers (called from ekf_api.hip)
// ---------------------------------------------------------------------------------------------

long cadence_gbuf_doubles() { return (long)CAD_GP * CAD_ROWS * CAD_CS; }   // per trajectory: CAD_GP parts

// (look-ahead) the next cadence's block while `kb` ranks are pending -> gbuf
void launch_gather_cad(hipStream_t st, const double* P, const double* V, const double* W, const double* dacc,
                       const StepIn* in, const CadPlan* plan, int batch, int kb, const DeviceConfig& cfg, int ld, long pstride,
                       double* gbuf) {
  hipLaunchKernelGGL(k_gather_cad, dim3((kb + 7) / 8, batch), dim3(64 * CAD_GW), 0, st, P, V, W, dacc, in, plan, batch, kb, cfg,
                     ld, pstride, gbuf);
}

// `colbuf` (batch x CAD_CU x ld doubles, or nullptr): the launch also gathers the mirrored column entries of the panel launch
// behind it, on `col_wgs` extra workgroups -- only where P_base is current (not beside a pass: look-ahead)
// `chain`: the instantiation that also records the pose block behind the cadence (CadOut::posefin) for k_chain_cad; `gmu`
// (with gbuf, one part): block and mean come from k_chain_cad
void launch_solve_cad(hipStream_t st, const double* P, const double* mu_in, double* mu_out, double* dacc_out,
                      const int* nact, const StepIn* in, const CadPlan* plan, int batch, CadOut* out, unsigned* flags,
                      const DeviceConfig& cfg, int ld, long pstride, const double* gbuf, int gparts, double* colbuf, int n_hi,
                      int col_wgs, bool chain, const double* gmu, unsigned* sync, unsigned start_sigma, const CadPre* pre) {
  if (chain)
    hipLaunchKernelGGL(k_solve_cad<true>, dim3(batch + (colbuf ? col_wgs : 0)), dim3(64 * CAD_NW), 0, st, P, mu_in, mu_out, dacc_out,
                       nact, in, plan, batch, out, flags, cfg, ld, pstride, gbuf, gparts, colbuf, col_wgs, n_hi, gmu, sync, start_sigma,
                       pre);
  else
    hipLaunchKernelGGL(k_solve_cad<false>, dim3(batch + (colbuf ? col_wgs : 0)), dim3(64 * CAD_NW), 0, st, P, mu_in, mu_out, dacc_out,
                       nact, in, plan, batch, out, flags, cfg, ld, pstride, gbuf, gparts, colbuf, col_wgs, n_hi, nullptr, nullptr, 0u,
                       nullptr);
}

// (chained runs) the next cadence's block and mean from the records `prev` of the cadence whose solve has just run
// gather workgroups per trajectory: each counts itself off on the launch's gather counter.  (Every workgroup of the launch
// reserves the chain workgroup's LDS, a CU apiece: as many as leave the chip half free for what runs beside the launch.)
int chain_gather_workgroups(int batch, int cus) { return std::max(1, std::min(CH_GW, (cus / 2 - batch) / std::max(batch, 1))); }
int chain_sync_words() { return SYNC_WORDS; }
void launch_chain_cad(hipStream_t st, const double* P, const double* prow3, const double* mu_land, const double* mu_pose,
                      const CadOut* prev, const StepIn* in, const CadPlan* plan, int batch, const DeviceConfig& cfg, int ld,
                      long pstride, double* gbuf, double* gmu, double* xg, double* bg, unsigned* sync, unsigned gather_target,
                      unsigned* flags, int gw, unsigned sigma, const CadPre* pre_in, CadPre* pre_out, const CadPlan* plan2,
                      bool wait_pass) {
  // (workgroups: the chain workgroup of every trajectory, its gather workgroups, and -- where a cadence follows the next one --
  //  the positions workgroup that forms that cadence's inputs ahead)
  hipLaunchKernelGGL(k_chain_cad, dim3(batch * (1 + gw + (pre_out ? 1 : 0))), dim3(64 * CAD_NW), 0, st, P, prow3, mu_land, mu_pose,
                     prev, in, plan, batch, cfg, ld, pstride, gbuf, gmu, xg, bg, sync, gather_target, flags, gw, sigma, pre_in, pre_out,
                     plan2, wait_pass ? 1 : 0);
}

void launch_mark(hipStream_t st, unsigned* sync, unsigned sigma) { hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, st, sync, sigma); }

void launch_gate(hipStream_t st, unsigned* sync, unsigned sigma, unsigned* flags, int batch) {
  hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, st, sync, sigma, flags, batch);
}
// workgroups of the panel launch (what decides between the launch being its own gate and the gate launch)
bool panels_cad_latency_regime(int batch, int n_hi) { return (long)((n_hi + 63) / 64) * batch <= CAD_KS_WAVES; }
int panels_cad_workgroups(int batch, int n_hi) {
  const long waves = (long)((n_hi + 63) / 64) * batch;
  return (int)(waves <= 1024 ? waves : (long)((n_hi + 255) / 256) * batch);
}

void launch_snap_pose(hipStream_t st, const double* P, const int* nact, int ld, long pstride, int batch, int n_hi, double* prow3) {
  hipLaunchKernelGGL(k_snap_pose, dim3((n_hi + 255) / 256, batch), dim3(256), 0, st, P, nact, ld, pstride, prow3);
}

// `nrp`: the ranks the bank's busiest trajectory appends, padded to a whole k-tile (every trajectory writes that many)
void launch_panels_cad(hipStream_t st, double* P, double* V, double* W, const double* mu_in, double* mu_out,
                       const int* nact, const CadOut* co, SolveOut* so, unsigned* queue, int ld, long pstride, int batch,
                       int n_hi, int nrp, const double* colbuf, double* prow3, unsigned* sync, unsigned head_sigma, unsigned tail_target,
                       unsigned* flags, bool tform, unsigned start_sigma, int shape, bool skipw) {
  // (`skipw`: only the replay shapes 2 and 3 honour it -- the caller asks for it only where the launch takes one of them)
  // (chained cadences in the latency regime: the triangular-solve form, one 8-wave workgroup per 64 state indices)
  if (tform) {
    hipLaunchKernelGGL(k_panels_cad_tf, dim3((n_hi + 63) / 64, batch), dim3(64 * CAD_NW), 0, st, P, V, W, mu_in, mu_out, nact, co, so,
                       queue, ld, pstride, nrp, prow3, sync, head_sigma, tail_target, flags, start_sigma);
    return;
  }
  // few state indices (the latency regime): four waves split the rows of the panel of 64 state indices (k_panels_cad_ks);
  // up to one wave per SIMD: one wave per workgroup
  // (`shape`: 0 = by the size of the launch; 1 .. 3 force a shape -- diagnostics, option "panel_shape")
  const long waves = (long)((n_hi + 63) / 64) * batch;
  if (shape == 0) shape = waves <= CAD_KS_WAVES ? 1 : (waves <= 1024 ? 2 : 3);
  if (shape == 1)
    hipLaunchKernelGGL(k_panels_cad_ks, dim3((n_hi + 63) / 64, batch), dim3(256), 0, st, P, V, W, mu_in, mu_out,
                       nact, co, so, queue, ld, pstride, nrp, colbuf, prow3, sync, head_sigma, tail_target, flags, start_sigma);
  else if (shape == 2)
    hipLaunchKernelGGL((k_panels_cad<1>), dim3((n_hi + 63) / 64, batch), dim3(64), 0, st, P, V, W, mu_in, mu_out,
                       nact, co, so, queue, ld, pstride, nrp, colbuf, prow3, sync, head_sigma, tail_target, flags, start_sigma, skipw ? 1 : 0);
  else
    hipLaunchKernelGGL((k_panels_cad<4>), dim3((n_hi + 255) / 256, batch), dim3(256), 0, st, P, V, W, mu_in,
                       mu_out, nact, co, so, queue, ld, pstride, nrp, colbuf, prow3, sync, head_sigma, tail_target, flags, start_sigma, skipw ? 1 : 0);
}

}  // namespace ekf
