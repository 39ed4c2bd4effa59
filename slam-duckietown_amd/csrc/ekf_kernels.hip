// Hand-written HIP kernels of the EKF-SLAM step for MI355X (gfx950, wave64).
//
// One reference step (src/replay_no_ros.py:363-482: predict, then one dense (I-KH)P product per
// observed landmark) is executed as
//   k_solve    sequential part on the compressed c x c system (c = 3+2m), one wave per trajectory
//   k_panels   V = T P'[C,:] (2m x n, coalesced row reads) and W = -P'[:,C] U (n x 2m), mean update
//   k_pass     P <- P + Rt + W V   one streaming read-modify-write of P (HBM-bound, 16 n^2 bytes)
// or, with no observation, k_predict_rc (rows/cols 0,1 of P only, O(n)).
// The algebra is restated on the CPU in oracle/ekf_oracle.py::ekf_step_structured.
#include "ekf_device.h"

namespace ekf {

__device__ __forceinline__ double wrap_pi(double a) {
  // (a + pi) % (2 pi) - pi with NumPy remainder semantics (src/replay_no_ros.py:397, :458)
  const double two_pi = 2.0 * M_PI;
  double r = fmod(a + M_PI, two_pi);
  if (r != 0.0) {
    if (r < 0.0) r += two_pi;
  } else {
    r = 0.0;
  }
  return r - M_PI;
}

// ---------------------------------------------------------------------------------------------
// k_solve: one 64-lane wave per trajectory.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_solve(const double* __restrict__ P, double* __restrict__ mu,
                                              const int* __restrict__ nact,
                                              const StepIn* __restrict__ in,
                                              SolveOut* __restrict__ out,
                                              unsigned* __restrict__ flags, DeviceConfig cfg, int ld,
                                              long pstride, int mcap) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const StepIn& s = in[b];
  SolveOut& o = out[b];
  const double* Pb = P + (long)b * pstride;
  double* mub = mu + (long)b * ld;

  __shared__ double Pc[CMAX][CMAX + 1], A[CMAX][CMAX + 1], Bm[CMAX][CMAX + 1];
  __shared__ double muc[CMAX];
  __shared__ double hp[2][CMAX], tj[2][CMAX], ph[CMAX][2], bh[CMAX][2], kc[CMAX][2], uj[CMAX][2];
  __shared__ int Cs[CMAX + 1];

  const bool do_pred = (s.flags & FLAG_PREDICT) != 0;
  int m = ((s.flags & FLAG_UPDATE) && cfg.enable_measurement_model) ? s.m : 0;
  if (m > MMAX) m = MMAX;
  const int c = 3 + 2 * m;

  if (lane < CMAX + 1) {
    int v = 0;
    if (lane < 3) v = lane;
    else if (lane < c) v = 3 + 2 * s.idx[(lane - 3) >> 1] + ((lane - 3) & 1);
    Cs[lane] = v;
    o.C[lane] = v;
  }
  __syncthreads();
  for (int e = lane; e < c * c; e += 64) {
    const int r = e / c, cc = e - r * c;
    Pc[r][cc] = Pb[(long)Cs[r] * ld + Cs[cc]];
    A[r][cc] = (r == cc) ? 1.0 : 0.0;
    Bm[r][cc] = (r == cc) ? 1.0 : 0.0;
  }
  if (lane < c) muc[lane] = mub[Cs[lane]];
  __syncthreads();

  // ---- motion model (src/replay_no_ros.py:368-417), evaluated redundantly by every lane ----
  const double th = muc[2];
  double g0 = 0.0, g1 = 0.0, nx = muc[0], ny = muc[1], nth = th;
  if (do_pred && !cfg.disable_motion_model) {
    const double lin = s.lin, ang = s.ang;
    if (cfg.enable_circular_interpolation) {
      if (fabs(ang) <= cfg.arc_threshold) {               // :376 straight, theta not advanced
        nx += lin * cos(th);
        ny += lin * sin(th);
        g0 = -lin * sin(th);
        g1 = lin * cos(th);
      } else {                                            // :390 arc
        const double r = lin / ang;
        nx += -r * sin(th) + r * sin(th + ang);
        ny += r * cos(th) - r * cos(th + ang);
        nth = wrap_pi(th + ang);                          // :397
        g0 = -r * cos(th) + r * cos(th + ang);            // :401
        g1 = -r * sin(th) + r * sin(th + ang);            // :402
      }
    } else {                                              // :405-417, no wrap
      nx += lin * cos(th);
      ny += lin * sin(th);
      nth = th + ang;
      g0 = -lin * sin(th);
      g1 = lin * cos(th);
    }
  }
  const double rd0 = do_pred ? cfg.rd[0] : 0.0, rd1 = do_pred ? cfg.rd[1] : 0.0,
               rd2 = do_pred ? cfg.rd[2] : 0.0;
  const double p22 = Pc[2][2];
  __syncthreads();
  // P'[C,C] = Gc P[C,C] Gc^T + Rt  (:428-430 restricted to C)
  if (lane < c) {
    const double r2 = Pc[2][lane];
    Pc[0][lane] += g0 * r2;
    Pc[1][lane] += g1 * r2;
  }
  __syncthreads();
  if (lane < c) {
    const double c2 = Pc[lane][2];
    Pc[lane][0] += g0 * c2;
    Pc[lane][1] += g1 * c2;
  }
  __syncthreads();
  if (lane == 0) {
    Pc[0][0] += rd0;
    Pc[1][1] += rd1;
    Pc[2][2] += rd2;
    muc[0] = nx;
    muc[1] = ny;
    muc[2] = nth;
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

  // ---- sequential per-landmark recurrences (:436-480) on the compressed system ----
  for (int j = 0; j < m; ++j) {
    const int a = 3 + 2 * j;
    const int sel[5] = {0, 1, 2, a, a + 1};
    const double dx = muc[a] - muc[0], dy = muc[a + 1] - muc[1];     // :443
    const double q = dx * dx + dy * dy;                               // :446
    const double sq = sqrt(q);
    const double zh1 = atan2(dy, dx) - muc[2];                        // :453
    const double y0 = s.range[j] - sq;                                // :455
    const double y1 = wrap_pi(s.bearing[j] - zh1);                    // :458
    // :466-469, elementwise (array / q) like NumPy so q == 0 propagates NaN/inf the same way
    const double h5[2][5] = {{(-sq * dx) / q, (-sq * dy) / q, 0.0 / q, (sq * dx) / q, (sq * dy) / q},
                             {dy / q, -dx / q, -q / q, -dy / q, dx / q}};
    if (lane < c) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        double s_hp = 0.0, s_ph = 0.0, s_t = 0.0, s_b = 0.0;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          s_hp += h5[r][k] * Pc[sel[k]][lane];      // (H P)[r, C[lane]]
          s_ph += Pc[lane][sel[k]] * h5[r][k];      // (P H^T)[C[lane], r]
          s_t += h5[r][k] * A[sel[k]][lane];
          s_b += Bm[lane][sel[k]] * h5[r][k];
        }
        hp[r][lane] = s_hp;
        ph[lane][r] = s_ph;
        tj[r][lane] = s_t;
        bh[lane][r] = s_b;
      }
    }
    __syncthreads();
    double S00 = 0.0, S01 = 0.0, S10 = 0.0, S11 = 0.0;                // :473  S = H P H^T + Q
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      S00 += hp[0][sel[k]] * h5[0][k];
      S01 += hp[0][sel[k]] * h5[1][k];
      S10 += hp[1][sel[k]] * h5[0][k];
      S11 += hp[1][sel[k]] * h5[1][k];
    }
    S00 += cfg.qd[0];
    S11 += cfg.qd[1];
    const double det = S00 * S11 - S01 * S10;
    const double i00 = S11 / det, i01 = -S01 / det, i10 = -S10 / det, i11 = S00 / det;
    if (lane < c) {
      const double k0 = ph[lane][0] * i00 + ph[lane][1] * i10;        // K_j[C[lane], :]
      const double k1 = ph[lane][0] * i01 + ph[lane][1] * i11;
      const double u0 = bh[lane][0] * i00 + bh[lane][1] * i10;
      const double u1 = bh[lane][0] * i01 + bh[lane][1] * i11;
      kc[lane][0] = k0;
      kc[lane][1] = k1;
      uj[lane][0] = u0;
      uj[lane][1] = u1;
      muc[lane] += k0 * y0 + k1 * y1;                                 // :476
      o.U[lane][2 * j] = u0;
      o.U[lane][2 * j + 1] = u1;
      o.T[2 * j][lane] = tj[0][lane];
      o.T[2 * j + 1][lane] = tj[1][lane];
    }
    if (lane == 0) {
      o.ys[2 * j] = y0;
      o.ys[2 * j + 1] = y1;
    }
    __syncthreads();
    for (int e = lane; e < c * c; e += 64) {                          // :480 restricted to C
      const int r = e / c, cc = e - r * c;
      Pc[r][cc] -= kc[r][0] * hp[0][cc] + kc[r][1] * hp[1][cc];
      A[r][cc] -= kc[r][0] * tj[0][cc] + kc[r][1] * tj[1][cc];
      Bm[r][cc] -= uj[r][0] * hp[0][cc] + uj[r][1] * hp[1][cc];
    }
    __syncthreads();
  }

  // zero the padding the templated consumers (cap = mcap landmarks) will read; each element once
  const int cc_cap = 3 + 2 * mcap, k_cap = 2 * mcap;
  for (int e = lane; e < k_cap * cc_cap; e += 64) {
    const int k = e / cc_cap, a = e - k * cc_cap;
    if (k >= 2 * m || a >= c) o.T[k][a] = 0.0;
    if (k >= 2 * m || a >= c) o.U[a][k] = 0.0;
  }
  if (lane < k_cap && lane >= 2 * m) o.ys[lane] = 0.0;
  bool bad = false;
  if (lane < c) {
    const double v = muc[lane];
    o.mu_c[lane] = v;
    mub[Cs[lane]] = v;
    bad = !(fabs(v) <= 1.79769313486231570815e308);
  }
  if (__any(bad) && lane == 0) atomicOr(&flags[b], EKF_FLAG_NONFINITE);
}

// ---------------------------------------------------------------------------------------------
// k_panels: thread j builds column j of V and row j of W.
//   V[k][j] = sum_a T[k][a] P'[C[a]][j]        k < 2m      (coalesced row reads)
//   W[j][k] = -sum_a P'[j][C[a]] U[a][k]                    (scattered column reads, ~m+1 lines/row)
//   rank-2 rows for the motion Jacobian: V[2m] = P[2,:]+p22h*gt, W[:,2m] = gt,
//                                        V[2m+1] = gt,           W[:,2m+1] = P[:,2]+p22h*gt
//   mean: mu[j] += Kst[j,:] . ys  for j not in C (k_solve already wrote mu[C]).
// ---------------------------------------------------------------------------------------------
template <int MCAP>
__global__ __launch_bounds__(64) void k_panels(const double* __restrict__ P, double* __restrict__ mu,
                                               const int* __restrict__ nact,
                                               const SolveOut* __restrict__ so,
                                               double* __restrict__ V, double* __restrict__ W, int ld,
                                               long pstride) {
  constexpr int CC = 3 + 2 * MCAP, K2 = 2 * MCAP, KT = K2 + 2;
  const int b = blockIdx.y;
  const int n = nact[b];
  if ((int)blockIdx.x * 64 >= n) return;
  const int j = blockIdx.x * 64 + threadIdx.x;
  const bool act = j < n;
  const int jj = act ? j : 0;
  const SolveOut& o = so[b];
  const double* Pb = P + (long)b * pstride;
  const double g0 = o.g[0], g1 = o.g[1];
  const double gj = (jj == 0) ? g0 : ((jj == 1) ? g1 : 0.0);

  double R[CC], L[CC];
#pragma unroll
  for (int a = 0; a < CC; ++a) R[a] = Pb[(long)o.C[a] * ld + jj];
#pragma unroll
  for (int a = 0; a < CC; ++a) L[a] = Pb[(long)jj * ld + o.C[a]];
  const double raw_r2 = R[2], raw_c2 = L[2];

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

  double* Vb = V + (long)b * (2 * MMAX + 2) * ld;
#pragma unroll
  for (int k = 0; k < K2; ++k) {
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < CC; ++a) acc += o.T[k][a] * R[a];
    if (act) Vb[(long)k * ld + j] = acc;
  }
  double dm = 0.0;
  double* Wr = W + ((long)b * ld + jj) * (2 * MMAX + 2);
#pragma unroll
  for (int k = 0; k < K2; ++k) {
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < CC; ++a) acc += L[a] * o.U[a][k];
    dm += acc * o.ys[k];
    if (act) Wr[k] = -acc;
  }
  if (act) {
    Vb[(long)K2 * ld + j] = raw_r2 + o.p22h * gj;
    Vb[(long)(K2 + 1) * ld + j] = gj;
    Wr[K2] = gj;
    Wr[K2 + 1] = raw_c2 + o.p22h * gj;
    bool inC = false;
    const int c = o.c;
    for (int a = 0; a < c; ++a) inC |= (o.C[a] == j);
    if (!inC) mu[(long)b * ld + j] += dm;
  }
  (void)KT;
}

// ---------------------------------------------------------------------------------------------
// k_pass: P[i][j] += Rt + sum_{k<KT} W[i][k] V[k][j], in place, one read + one write of P.
// A wave owns a strip of 128 columns (2 adjacent doubles per lane = one 1 KiB row segment per
// load/store instruction); its V strip lives in registers for the whole row block, W[i][:] is
// wave-uniform and comes through the scalar cache.  UNR rows are in flight per wave.
// ---------------------------------------------------------------------------------------------
template <int MCAP, int UNR>
__global__ __launch_bounds__(256) void k_pass(double* __restrict__ P, const double* __restrict__ V,
                                              const double* __restrict__ W,
                                              const int* __restrict__ nact,
                                              const SolveOut* __restrict__ so, int ld, long pstride,
                                              int rows_per_block) {
  constexpr int KT = 2 * MCAP + 2;
  constexpr int WS = 2 * MMAX + 2;
  const int b = blockIdx.z;
  const int n = nact[b];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int strip = blockIdx.x * 4 + wave;
  const int i0 = blockIdx.y * rows_per_block;
  if (strip * 128 >= n || i0 >= n) return;
  const int i1 = min(n, i0 + rows_per_block);
  const int j0 = strip * 128 + lane * 2;
  if (j0 >= n) return;
  const bool two = (j0 + 1) < n;

  double* Pb = P + (long)b * pstride;
  const double* Vb = V + (long)b * WS * ld;
  const double* Wb = W + (long)b * ld * WS;

  double2 v[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) v[k] = *reinterpret_cast<const double2*>(Vb + (long)k * ld + j0);

  int i = i0;
  for (; i + UNR <= i1; i += UNR) {
    double2 p[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) p[u] = *reinterpret_cast<const double2*>(Pb + (long)(i + u) * ld + j0);
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
    for (int u = 0; u < UNR; ++u) {
      double* dst = Pb + (long)(i + u) * ld + j0;
      if (two) *reinterpret_cast<double2*>(dst) = p[u];
      else *dst = p[u].x;
    }
  }
  for (; i < i1; ++i) {
    double2 p = *reinterpret_cast<const double2*>(Pb + (long)i * ld + j0);
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
    double* dst = Pb + (long)i * ld + j0;
    if (two) *reinterpret_cast<double2*>(dst) = p;
    else *dst = p.x;
  }
}

// ---------------------------------------------------------------------------------------------
// k_predict_rc: prediction with no observation touches only rows/cols 0,1 and the pose diagonal
// (src/replay_no_ros.py:428-430 with G_F = I outside the 3x3 block).  O(n).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_predict_rc(double* __restrict__ P,
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
void launch_solve(hipStream_t st, const double* P, double* mu, const int* nact, const StepIn* in,
                  SolveOut* out, unsigned* flags, const DeviceConfig& cfg, int ld, long pstride,
                  int batch, int mcap) {
  hipLaunchKernelGGL(k_solve, dim3(batch), dim3(64), 0, st, P, mu, nact, in, out, flags, cfg, ld,
                     pstride, mcap);
}

template <int MCAP>
static void launch_panels_t(hipStream_t st, const double* P, double* mu, const int* nact,
                            const SolveOut* so, double* V, double* W, int ld, long pstride, int batch,
                            int n_hi) {
  hipLaunchKernelGGL(k_panels<MCAP>, dim3((n_hi + 63) / 64, batch), dim3(64), 0, st, P, mu, nact, so,
                     V, W, ld, pstride);
}

void launch_panels(hipStream_t st, int mcap, const double* P, double* mu, const int* nact,
                   const SolveOut* so, double* V, double* W, int ld, long pstride, int batch,
                   int n_hi) {
  switch (mcap) {
    case 1: launch_panels_t<1>(st, P, mu, nact, so, V, W, ld, pstride, batch, n_hi); break;
    case 2: launch_panels_t<2>(st, P, mu, nact, so, V, W, ld, pstride, batch, n_hi); break;
    case 4: launch_panels_t<4>(st, P, mu, nact, so, V, W, ld, pstride, batch, n_hi); break;
    case 8: launch_panels_t<8>(st, P, mu, nact, so, V, W, ld, pstride, batch, n_hi); break;
    default: launch_panels_t<16>(st, P, mu, nact, so, V, W, ld, pstride, batch, n_hi); break;
  }
}

template <int MCAP, int UNR>
static void launch_pass_t(hipStream_t st, double* P, const double* V, const double* W,
                          const int* nact, const SolveOut* so, int ld, long pstride, int batch,
                          int n_hi, int rows_per_block) {
  dim3 grid((n_hi + 511) / 512, (n_hi + rows_per_block - 1) / rows_per_block, batch);
  hipLaunchKernelGGL((k_pass<MCAP, UNR>), grid, dim3(256), 0, st, P, V, W, nact, so, ld, pstride,
                     rows_per_block);
}

void launch_pass(hipStream_t st, int mcap, double* P, const double* V, const double* W,
                 const int* nact, const SolveOut* so, int ld, long pstride, int batch, int n_hi,
                 int rows_per_block) {
  switch (mcap) {
    case 1: launch_pass_t<1, 8>(st, P, V, W, nact, so, ld, pstride, batch, n_hi, rows_per_block); break;
    case 2: launch_pass_t<2, 8>(st, P, V, W, nact, so, ld, pstride, batch, n_hi, rows_per_block); break;
    case 4: launch_pass_t<4, 8>(st, P, V, W, nact, so, ld, pstride, batch, n_hi, rows_per_block); break;
    case 8: launch_pass_t<8, 8>(st, P, V, W, nact, so, ld, pstride, batch, n_hi, rows_per_block); break;
    default: launch_pass_t<16, 4>(st, P, V, W, nact, so, ld, pstride, batch, n_hi, rows_per_block); break;
  }
}

void launch_predict_rc(hipStream_t st, double* P, const int* nact, const SolveOut* so, int ld,
                       long pstride, int batch, int n_hi) {
  hipLaunchKernelGGL(k_predict_rc, dim3((n_hi + 255) / 256, batch), dim3(256), 0, st, P, nact, so, ld,
                     pstride);
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
