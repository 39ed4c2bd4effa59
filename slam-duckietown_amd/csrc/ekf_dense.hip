// General dense covariance propagation  P <- F P F^T + Q  on the fp64 matrix cores of gfx950.
//
// This is the reference's literal product G_F @ P @ G_F.T + F.T @ R @ F (src/replay_no_ros.py:430)
// for an arbitrary n x n Jacobian F: two GEMMs of n^3 MACs each (tmp = F P, P = tmp F^T + Q).
// v_mfma_f64_16x16x4_f64: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15] one double per lane;
// C/D four doubles per lane, col = l&15, row = (l>>4) + 4*reg  (the f64 map, not the f32 one).
//
// Workgroup = 4 waves, block tile 128 x 128, K slab 16, each wave a 64 x 64 quadrant = 4x4 MFMA
// tiles (64 accumulator doubles per lane).  Both operands are staged k-major in LDS
// (tile[k][row], row stride 144 doubles so the four k-groups of a ds_read_b64 hit disjoint banks);
// the next slab is prefetched into registers while the current one feeds the MFMAs.
#include "ekf_device.h"

namespace ekf {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16, LS = 144;

// Load 8 consecutive k-entries of one row (row-major source, k contiguous): rows >= n or k >= n -> 0.
__device__ __forceinline__ void load_row8(const double* __restrict__ src, int ld, int n, int row, int k0,
                                          double (&r)[8]) {
  if (row < n && k0 + 8 <= n) {                        // interior: four unconditional 16-byte loads
    const double2* p = reinterpret_cast<const double2*>(src + (long)row * ld + k0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double2 t = p[q];
      r[2 * q] = t.x;
      r[2 * q + 1] = t.y;
    }
  } else if (row < n) {
    const double2* p = reinterpret_cast<const double2*>(src + (long)row * ld + k0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double2 t = (k0 + 2 * q < n) ? p[q] : make_double2(0.0, 0.0);
      r[2 * q] = t.x;
      r[2 * q + 1] = (k0 + 2 * q + 1 < n) ? t.y : 0.0;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) r[q] = 0.0;
  }
}

// TRANS_B = false: C = A * B     (B row-major [k][j])
// TRANS_B = true : C = A * B^T + Q  (B row-major [j][k])
template <bool TRANS_B>
__global__ __launch_bounds__(256, 2) void k_gemm_f64(const double* __restrict__ A,
                                                  const double* __restrict__ B,
                                                  const double* __restrict__ Q, double* __restrict__ C,
                                                  int n, int ld) {
  __shared__ double As[2][BK][LS];
  __shared__ double Bs[2][BK][LS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bm = blockIdx.y * BM, bn = blockIdx.x * BN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 15, lk = lane >> 4;

  double4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

  // staging assignment: "row panel" = 128 rows x 16 k (thread: row t>>1, k half (t&1)*8)
  //                     "k panel"   = 16 k x 128 cols (thread: k t>>4, cols (t&15)*8)
  const int rp_row = tid >> 1, rp_k = (tid & 1) * 8;
  const int kp_k = tid >> 4, kp_c = (tid & 15) * 8;
  double ra[8], rb[8];

  auto fetch = [&](int k0) {
    load_row8(A, ld, n, bm + rp_row, k0 + rp_k, ra);
    if (TRANS_B) {
      load_row8(B, ld, n, bn + rp_row, k0 + rp_k, rb);
    } else {
      // row k0+kp_k of B, columns bn+kp_c .. +8
      load_row8(B, ld, n, k0 + kp_k, bn + kp_c, rb);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 8; ++q) As[buf][rp_k + q][rp_row] = ra[q];
    if (TRANS_B) {
#pragma unroll
      for (int q = 0; q < 8; ++q) Bs[buf][rp_k + q][rp_row] = rb[q];
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) Bs[buf][kp_k][kp_c + q] = rb[q];
    }
  };

  const int nk = (n + BK - 1) / BK;
  fetch(0);
  stash(0);
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    if (t + 1 < nk) fetch((t + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[buf][kk + lk][wm + i * 16 + li];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[buf][kk + lk][wn + j * 16 + li];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nk) stash(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = bn + wn + j * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = bm + wm + i * 16 + lk + 4 * r;
        if (row < n && col < n) {
          double v = acc[i][j][r];
          if (TRANS_B) v += Q[(long)row * ld + col];
          C[(long)row * ld + col] = v;
        }
      }
    }
}

int dense_propagate(hipStream_t st, double* P, double* tmp, const double* F, const double* Q, int n, int ld) {
  dim3 grid((n + BN - 1) / BN, (n + BM - 1) / BM);
  hipLaunchKernelGGL(k_gemm_f64<false>, grid, dim3(256), 0, st, F, P, nullptr, tmp, n, ld);   // tmp = F P
  hipLaunchKernelGGL(k_gemm_f64<true>, grid, dim3(256), 0, st, tmp, F, Q, P, n, ld);          // P = tmp F^T + Q
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace ekf
