// The SMALL-STATE path (gfx950, wave64): a small filter -- n <= SMALL_N_MAX = 79, i.e. up to 38 landmarks (n <= 131, 64 landmarks,
// in a bank of at least 128 trajectories): the reference's real map has 12 (src/replay_no_ros.py:26) -- runs every step, or a whole uploaded stream of steps, inside ONE workgroup with P
// resident in LDS.  (The limit is where this path stops winning, not where LDS ends: tools/step_latency.py, N = 38: 25.7 us per
// online step and 13.0 us per streamed step against 27.5 / 18.9 us on the general kernels; N = 45: 30.6 / 16.7 against 27.6 / 19.0;
// N = 64: 44 / 25 against 28 / 19 -- the down-date and the per-launch load / store of the triangle grow with n^2.)
//
// The general kernels are built for covariances that live in HBM: a step appends ranks, the O(n^2) pass is deferred, and every
// step pays a sequential "solve" chain of ~1.4 us per landmark on the compressed block plus two or three dependent launches.
// At n = 43 (N = 20) all of P is 14.8 KB: nothing needs deferring.  Here a 256-thread workgroup per trajectory
//   * loads the stored upper triangle of P_base into LDS (mirrored: the matrix is kept exactly symmetric, every update
//     computes an entry (a, b), a <= b, once and writes it to both places -- the device's "upper triangle is authoritative"),
//   * for every step of the launch: motion model, P <- G P G^T + R on rows / columns 0, 1 (src/replay_no_ros.py:368-430),
//     then the landmarks in order (:436-480): Jacobian at the current mean, (H P) by thread = column, S = H P H^T + Q and its
//     closed-form inverse, K = (H P)^T S^-1 (P symmetric), mean += K y, P <- P - K (H P) on the upper triangle -- the SAME
//     simple-form, sequentially re-linearised arithmetic as the reference, three workgroup barriers per landmark,
//   * writes mean and upper triangle back.
// One launch per online step (instead of solve + panels [+ pass]); ONE launch for a whole uploaded stream.  N = 20, m = 8, one
// trajectory: 9.8 us per step against 18 - 19 us on the general path; a bank of small filters is one workgroup each.
// No rank is ever pending on this path (pending_k stays 0), so every other entry point -- uploads, downloads, augmentation,
// device-side association, the dense product -- works on P_base as it stands.
#include <atomic>
#include "ekf_devfn.h"

namespace ekf {

constexpr int SMALL_N_MAX = 79;         // 3 + 2 * 38 (<= 5 column tiles of 16; 50 KB of LDS): beyond, the general kernels are faster
                                        // for ONE trajectory (latency) ...
constexpr int SMALL_N_MAX_BANK = 131;   // ... but not for a bank that fills the chip (3 + 2 * 64; 9 column tiles, 137 KB of LDS: one
constexpr int SMALL_BANK_MIN = 128;     // workgroup per CU): N = 64 x 256 10.4 M against 7.9 M steps/s, x 1024 against 5.4 M

// One step's landmark updates and prediction on the LDS-resident state.  `Pl` is n x ps (ps odd: row and column walks are
// both conflict-free), `mu` the mean, `hp` / `kk` 2 x n scratch.
template <int NT, int TM>
__device__ __forceinline__ void small_step(double* __restrict__ Pl, double* __restrict__ mu, double* __restrict__ hp,
                                           double* __restrict__ kk, double* __restrict__ sc, const StepIn& s,
                                           const DeviceConfig& cfg, int n, int ps) {
  const int tid = threadIdx.x;
  const bool do_pred = (s.flags & FLAG_PREDICT) != 0;
  int m = ((s.flags & FLAG_UPDATE) && cfg.enable_measurement_model) ? s.m : 0;
  m = min(m, MMAX);
  const int n_lm = (n - 3) >> 1;
  // ---- prediction (:368-430) ----
  if (do_pred) {
    if (tid == 0) {
      const double th = mu[2];
      double g0 = 0.0, g1 = 0.0, nx = mu[0], ny = mu[1], nth = th;
      if (!cfg.disable_motion_model) {
        const double lin = s.lin, ang = s.ang;
        double s0, c0;
        sincos(th, &s0, &c0);
        if (cfg.enable_circular_interpolation && fabs(ang) > cfg.arc_threshold) {   // :390 arc
          double s1, c1;
          sincos(th + ang, &s1, &c1);
          const double r = lin / ang;
          nx += -r * s0 + r * s1;
          ny += r * c0 - r * c1;
          nth = wrap_pi(th + ang);                          // :397
          g0 = -r * c0 + r * c1;                            // :401
          g1 = -r * s0 + r * s1;                            // :402
        } else {                                            // :376 straight / :405-417 linear mode
          nx += lin * c0;
          ny += lin * s0;
          if (!cfg.enable_circular_interpolation) nth = th + ang;   // no wrap (:409); :381 keeps theta
          g0 = -lin * s0;
          g1 = lin * c0;
        }
      }
      mu[0] = nx;
      mu[1] = ny;
      mu[2] = nth;
      sc[0] = g0;
      sc[1] = g1;
    }
    __syncthreads();
    const double g0 = sc[0], g1 = sc[1];
    // G P: rows 0, 1 take g_r x row 2 (all columns); then (G P) G^T: columns 0, 1 take g_c x column 2 (all rows).  Done on the
    // upper triangle's representatives and mirrored: entries (0, j), (1, j) for j >= 2 are P(r, j) + g_r P(2, j); the 3 x 3 pose
    // block is formed by one thread from its six stored entries exactly as the dense product forms it.
    for (int j = 3 + tid; j < n; j += NT) {
      const double p2 = Pl[2 * ps + j];
      const double v0 = fma(g0, p2, Pl[0 * ps + j]), v1 = fma(g1, p2, Pl[1 * ps + j]);
      Pl[0 * ps + j] = v0;
      Pl[j * ps + 0] = v0;
      Pl[1 * ps + j] = v1;
      Pl[j * ps + 1] = v1;
    }
    if (tid == 0) {
      double X[3][3], Y[3][3];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) X[r][c] = Pl[r * ps + c];
      for (int c = 0; c < 3; ++c) {                          // rows 0, 1 of G P
        X[0][c] = fma(g0, X[2][c], X[0][c]);
        X[1][c] = fma(g1, X[2][c], X[1][c]);
      }
      for (int r = 0; r < 3; ++r) {                          // columns 0, 1 of (G P) G^T
        Y[r][0] = fma(g0, X[r][2], X[r][0]);
        Y[r][1] = fma(g1, X[r][2], X[r][1]);
        Y[r][2] = X[r][2];
      }
      for (int r = 0; r < 3; ++r) Y[r][r] += cfg.rd[r];      // + F^T R F (:421-430)
      for (int r = 0; r < 3; ++r)
        for (int c = r; c < 3; ++c) {                        // the upper triangle is authoritative
          Pl[r * ps + c] = Y[r][c];
          Pl[c * ps + r] = Y[r][c];
        }
    }
    __syncthreads();
  }
  // ---- the landmarks, in order (:436-480) ----
  // What one landmark costs is latency, so every phase is as parallel as its data allow: the columns of (H P) one per thread
  // (n <= 79: waves 0 - 1), the innovation (atan2 + wrap: the longest scalar chain, needed by the mean only) on wave 3 beside
  // them, the rank-2 down-date as 16 x 16 tiles of the upper triangle over all 256 threads (independent iterations: the
  // loads of the next tiles are in flight under this one's FMAs).
  const int ty = tid >> 4, tx = tid & 15;
  for (int j = 0; j < m; ++j) {
    const int lm = s.idx[j];
    if (lm < 0 || lm >= n_lm) continue;                     // (uniform; validated on the host, and by k_associate)
    const int t = 3 + 2 * lm;
    double h[2][5];
    const LinGeom g = linearize_h(mu[0], mu[1], mu[2], mu[t], mu[t + 1], h);     // every thread: broadcast LDS reads
    if (tid >= NT - 64) {                                   // (wave 3) the innovation, under the other waves' (H P)
      double y0, y1;
#ifdef SM_SKIP_INNOVATION                                /* diagnostic build (timing only, wrong results): no atan2 / wrap */
      y0 = s.range[j] - g.sq;
      y1 = s.bearing[j] - g.th;
#else
      innovation(g, s.range[j], s.bearing[j], y0, y1);
#endif
      if (tid == NT - 64) {
        sc[2] = y0;
        sc[3] = y1;
      }
    }
    // (H P)[:, c] for this thread's column c: rows sel = {0, 1, 2, t, t + 1} of P
    for (int c = tid; c < n; c += NT) {
      double e0 = h[0][0] * Pl[c], e1 = h[1][0] * Pl[c];
#pragma unroll
      for (int k = 1; k < 5; ++k) {
        const double pv = Pl[((k < 3) ? k : t + (k - 3)) * ps + c];
        e0 = fma(h[0][k], pv, e0);
        e1 = fma(h[1][k], pv, e1);
      }
      hp[c] = e0;
      hp[n + c] = e1;
    }
    __syncthreads();
    // S = H P H^T + Q (:473), every thread redundantly from the five pairs at sel; closed-form inverse
    double S00 = cfg.qd[0], S01 = 0.0, S10 = 0.0, S11 = cfg.qd[1];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int c = (k < 3) ? k : t + (k - 3);
      const double a0 = hp[c], a1 = hp[n + c];
      S00 = fma(a0, h[0][k], S00);
      S01 = fma(a0, h[1][k], S01);
      S10 = fma(a1, h[0][k], S10);
      S11 = fma(a1, h[1][k], S11);
    }
#ifdef SM_FAST_RCP                                       /* diagnostic build (timing only): hardware reciprocal estimate */
    const double rdet = __builtin_amdgcn_rcp(S00 * S11 - S01 * S10);
#else
    const double rdet = 1.0 / (S00 * S11 - S01 * S10);
#endif
    const double i00 = S11 * rdet, i01 = -S01 * rdet, i10 = -S10 * rdet, i11 = S00 * rdet;
    for (int c = tid; c < n; c += NT) {                     // K[c, :] = (H P)[:, c]^T S^-1   (P symmetric)
      const double a0 = hp[c], a1 = hp[n + c];
      kk[c] = a0 * i00 + a1 * i10;
      kk[n + c] = a0 * i01 + a1 * i11;
    }
    __syncthreads();
    const double y0 = sc[2], y1 = sc[3];
    // P <- P - K (H P) on the upper triangle, mirrored (:480): 16 x 16 tiles (i <= j), thread (ty, tx) -> entry (16 i + ty, 16 j + tx)
    // (every load of a row of tiles is unconditional, at clamped addresses, and in flight before the first FMA; only the
    //  stores are predicated -- with the loads under the `a <= b` branch each tile was its own trip to LDS: 8 of 16 us per step)
#ifndef SM_SKIP_UPDATE                                   /* diagnostic build (timing only, wrong results): no covariance down-date */
    {
      // ALL tiles in flight at once -- one round trip to LDS for the whole down-date
      constexpr int NTILE = TM * (TM + 1) / 2;
      double pv[NTILE], h0[TM], h1[TM], k0[TM], k1[TM];
#pragma unroll
      for (int u = 0; u < TM; ++u) {
        const int bc = min(16 * u + tx, n - 1), ac = min(16 * u + ty, n - 1);
        h0[u] = hp[bc];
        h1[u] = hp[n + bc];
        k0[u] = kk[ac];
        k1[u] = kk[n + ac];
      }
#pragma unroll
      for (int i = 0, q = 0; i < TM; ++i)
#pragma unroll
        for (int u = i; u < TM; ++u, ++q) pv[q] = Pl[min(16 * i + ty, n - 1) * ps + min(16 * u + tx, n - 1)];
#pragma unroll
      for (int i = 0, q = 0; i < TM; ++i)
#pragma unroll
        for (int u = i; u < TM; ++u, ++q) {
          const int a = 16 * i + ty, b = 16 * u + tx;
          const double v = pv[q] - (k0[i] * h0[u] + k1[i] * h1[u]);
          if (a <= b && b < n) {
            Pl[a * ps + b] = v;
            Pl[b * ps + a] = v;
          }
        }
    }
#endif
    for (int c = tid; c < n; c += NT) mu[c] += kk[c] * y0 + kk[n + c] * y1;     // :476
    __syncthreads();
  }
}

// One workgroup per trajectory runs `nsteps` steps: in[k * batch + b], k = 0 .. nsteps - 1.
// TM: column tiles of 16 the state spans at most (n <= 16 TM): the down-date's loads are unrolled over them.
template <int NT, int TM>
__device__ __forceinline__ void small_stream_body(double* __restrict__ P, const double* __restrict__ mu_in,
                                                  double* __restrict__ mu_out, const int* __restrict__ nact,
                                                  const StepIn* __restrict__ in, int batch, int nsteps,
                                                  unsigned* __restrict__ flags, const DeviceConfig& cfg, int ld, long pstride,
                                                  double* __restrict__ host_out, int out_b,
                                                  unsigned long long* __restrict__ host_seq, unsigned long long out_seq) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = min(min(nact[b], SMALL_N_MAX_BANK), 16 * TM);
  constexpr int NQ = (16 * TM + 63) / 64;                   // columns per lane and row of the triangle's load / store
  const int ps = n | 1;                                     // odd row stride
  double* Pl = lds;
  double* mu = Pl + n * ps;
  double* hp = mu + n;
  double* kk = hp + 2 * n;
  double* sc = kk + 2 * n;                                  // 4 scalars
  // the step records are double-buffered in LDS: record k + 1 is fetched (one coalesced round trip, 44 words) while step k
  // runs -- read on demand from global memory every field of a record is its own dependent trip to L2 / HBM
  constexpr int RW = (int)(sizeof(StepIn) / 8);
  static_assert(sizeof(StepIn) % 8 == 0 && RW <= NT, "one 8-byte word of a record per thread");
  unsigned long long* recw = reinterpret_cast<unsigned long long*>(sc + 4);   // 2 x RW words
  double* Pb = P + (long)b * pstride;
  // the stored upper triangle, rows dealt to the waves, columns to the lanes (coalesced), four rows' loads in flight together;
  // every entry is written to both places of the LDS matrix (n <= 79 < 2 x 64: at most two columns per lane and row)
  {
    const int w = tid >> 6, lane = tid & 63;
    constexpr int NWV = NT / 64, U = 4;
    for (int r0 = w; r0 < n; r0 += NWV * U) {
      double v[U][3];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = min(r0 + NWV * u, n - 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int c = min(max(lane + 64 * q, r), n - 1);          // (clamped into the row's stored part: no load under a branch)
          v[u][q] = Pb[p_index(ld, r, c)];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = r0 + NWV * u;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int c = lane + 64 * q;
          if (r < n && c >= r && c < n) {
            Pl[r * ps + c] = v[u][q];
            Pl[c * ps + r] = v[u][q];
          }
        }
      }
    }
  }
  for (int c = tid; c < n; c += NT) mu[c] = mu_in[(long)b * ld + c];
  if (tid < RW) recw[tid] = reinterpret_cast<const unsigned long long*>(in + b)[tid];
  __syncthreads();
  for (int k = 0; k < nsteps; ++k) {
    unsigned long long nxt = 0;
    const bool more = k + 1 < nsteps;
    if (more && tid < RW) nxt = reinterpret_cast<const unsigned long long*>(in + (long)(k + 1) * batch + b)[tid];
    small_step<NT, TM>(Pl, mu, hp, kk, sc, *reinterpret_cast<const StepIn*>(recw + (k & 1) * RW), cfg, n, ps);
    if (more && tid < RW) recw[((k + 1) & 1) * RW + tid] = nxt;
    __syncthreads();
  }
  bool bad = false;
  for (int c = tid; c < n; c += NT) {
    const double v = mu[c];
    mu_out[(long)b * ld + c] = v;
    bad |= !(fabs(v) <= 1.79769313486231570815e308);
  }
  const bool any_bad = __syncthreads_or(bad);
  if (any_bad && tid == 0) atomicOr(flags + b, EKF_FLAG_NONFINITE);
  if (host_out && b == out_b) {
    // ekf_step_fetch: the state the caller asked for goes straight from LDS into pinned host memory, in k_pack_small's
    // layout (dense mirrored n x n covariance, mean, sticky flags) -- no second launch between the step and the host
    unsigned long long sum = 0ull;                      // XOR of the bit patterns this thread hands over
    for (int e = tid; e < n * n; e += NT) {
      const int r = e / n, c = e - r * n;
      const double v = Pl[r * ps + c];
      host_out[e] = v;
      sum ^= __builtin_bit_cast(unsigned long long, v);
    }
    for (int c = tid; c < n; c += NT) {
      host_out[n * n + c] = mu[c];
      sum ^= __builtin_bit_cast(unsigned long long, mu[c]);
    }
    if (tid == 0) {
      const double fl = (double)(atomicOr(flags + b, 0u) | (any_bad ? EKF_FLAG_NONFINITE : 0u));
      host_out[n * n + n] = fl;
      sum ^= __builtin_bit_cast(unsigned long long, fl);
    }
    // Integrity trailer behind the payload: the call's sequence number, written by the LAST thread (another wave than the
    // one that releases the flag word), and the XOR of every word of the payload (wave reduction, then one atomic per wave
    // into LDS).  The host compares the first always and the second when asked to ("fetch_verify"): see ekf_step_fetch.
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum ^= __shfl_xor(sum, o);
    __shared__ unsigned long long xsum;
    if (tid == 0) xsum = 0ull;
    __syncthreads();
    if ((tid & 63) == 0) atomicXor(&xsum, sum);
    __syncthreads();
    if (tid == NT - 1) {
      reinterpret_cast<unsigned long long*>(host_out)[n * n + n + 1] = out_seq;
      reinterpret_cast<unsigned long long*>(host_out)[n * n + n + 2] = xsum;
    }
    // the host does not wait for the launch to retire (completion signal, interrupt or poll of the runtime: 10 - 15 us) but
    // polls this word: every thread's stores are fenced to system scope, then the call's sequence number is released.
    // ASSUMPTION (validated on the MI355X boxes of this pool): posted writes of one kernel to
    // coherent pinned host memory become visible in the order fence -> release, i.e. PCIe relaxed ordering does not let the
    // flag word overtake the payload.  The trailer above is the tripwire for a platform where that does not hold.
    __threadfence_system();
    __syncthreads();
    if (tid == 0)
      __hip_atomic_store(host_seq, out_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  {
    const int w = tid >> 6, lane = tid & 63;
    for (int r = w; r < n; r += NT / 64)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int c = lane + 64 * q;
        if (c >= r && c < n) Pb[p_index(ld, r, c)] = Pl[r * ps + c];
      }
  }
}

#define SMALL_STREAM_ARGS                                                                                                  \
  double *__restrict__ P, const double *__restrict__ mu_in, double *__restrict__ mu_out, const int *__restrict__ nact,      \
      const StepIn *__restrict__ in, int batch, int nsteps, unsigned *__restrict__ flags, DeviceConfig cfg, int ld,        \
      long pstride, double *__restrict__ host_out, int out_b, unsigned long long *__restrict__ host_seq,                   \
      unsigned long long out_seq
#define SMALL_STREAM_PASS P, mu_in, mu_out, nact, in, batch, nsteps, flags, cfg, ld, pstride, host_out, out_b, host_seq, out_seq
// The latency form: whatever registers the compiler wants (148 / 203 VGPRs: three / two workgroups resident per CU) ...
template <int NT, int TM>
__global__ __launch_bounds__(NT) void k_small_stream(SMALL_STREAM_ARGS) {
  small_stream_body<NT, TM>(SMALL_STREAM_PASS);
}
// ... and the throughput form for banks that more than fill the chip at that occupancy: the same code held to 128 VGPRs (a few
// spilled registers: one trajectory alone is 10 % slower), four workgroups per CU -- banks of 1024 +34 %, of 4096 +14 %
// (tools/small_bank_sweep.py).  Same instructions on the data: bit-identical results.
template <int NT, int TM>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_small_stream_occ(SMALL_STREAM_ARGS) {
  small_stream_body<NT, TM>(SMALL_STREAM_PASS);
}
// ... and the 7-tile form (81 <= n <= 112, banks only) held to 256 VGPRs instead of 266: two workgroups per CU where LDS allows
// (n <= 93: 69 KB each).
template <int NT, int TM>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_small_stream_two(SMALL_STREAM_ARGS) {
  small_stream_body<NT, TM>(SMALL_STREAM_PASS);
}

int small_state_limit(int batch) { return batch >= SMALL_BANK_MIN ? SMALL_N_MAX_BANK : SMALL_N_MAX; }

int launch_small_stream(hipStream_t st, double* P, const double* mu_in, double* mu_out, const int* nact, const StepIn* in,
                        int batch, int nsteps, unsigned* flags, const DeviceConfig& cfg, int ld, long pstride, int n_hi,
                        double* host_out, int out_b, unsigned long long* host_seq, unsigned long long out_seq, bool many) {
  const int n = n_hi < SMALL_N_MAX_BANK ? n_hi : SMALL_N_MAX_BANK, ps = n | 1;
  const size_t bytes = sizeof(double) * ((size_t)n * ps + 5 * (size_t)n + 4) + 2 * sizeof(StepIn);
#define EKF_SMALL(K, TM)                                                                                                 \
  hipLaunchKernelGGL((K<256, TM>), dim3(batch), dim3(256), bytes, st, P, mu_in, mu_out, nact, in, batch, nsteps, flags, cfg, ld, \
                     pstride, host_out, out_b, host_seq, out_seq)
  // (more than 64 KB of dynamic LDS has to be asked for, once per kernel and device)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1;
#define EKF_SMALL_BIG(K, TM)                                                                                             \
  do {                                                                                                                   \
    static std::atomic<unsigned long long> asked{0};   /* (handles on several host threads: a lost bit would only repeat the call) */ \
    if (!((asked.load(std::memory_order_relaxed) >> dev) & 1ull)) {                                                      \
      const size_t most = sizeof(double) * ((size_t)SMALL_N_MAX_BANK * (SMALL_N_MAX_BANK | 1) + 5 * SMALL_N_MAX_BANK + 4) + \
                          2 * sizeof(StepIn);                                                                            \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&K<256, TM>), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                              (int)most) != hipSuccess)                                                                  \
        return 1;                                                                                                        \
      asked.fetch_or(1ull << dev, std::memory_order_relaxed);                                                            \
    }                                                                                                                    \
    EKF_SMALL(K, TM);                                                                                                    \
  } while (0)
  if (many && n <= 48) EKF_SMALL(k_small_stream_occ, 3);
  else if (n <= 48) EKF_SMALL(k_small_stream, 3);
  else if (n <= 80) EKF_SMALL(k_small_stream, 5);
  else if (n <= 112) EKF_SMALL_BIG(k_small_stream_two, 7);
  else EKF_SMALL_BIG(k_small_stream, 9);
#undef EKF_SMALL_BIG
#undef EKF_SMALL
  return 0;
}

}  // namespace ekf
