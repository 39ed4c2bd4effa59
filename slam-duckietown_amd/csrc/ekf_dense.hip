// General dense covariance propagation  P <- F P F^T + Q  on the fp64 matrix cores of gfx950.
//
// This is the reference's literal product G_F @ P @ G_F.T + F.T @ R @ F (src/replay_no_ros.py:430)
// for an arbitrary n x n Jacobian F: two GEMMs of n^3 MACs each (tmp = F P, P = tmp F^T + Q).
// v_mfma_f64_16x16x4_f64: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15] one double per lane;
// C/D four doubles per lane, col = l&15, row = (l>>4) + 4*reg  (the f64 map, not the f32 one).
//
// Workgroup = 4 waves, block tile 128 x 128, K slab 16, each wave a 64 x 64 quadrant = 4x4 MFMA
// tiles (64 accumulator doubles per lane).  Both operands are staged k-major in LDS
// (tile[k][row], row stride 144 doubles so the four k-groups of a ds_read_b64 hit disjoint banks).
#include <type_traits>

#include "ekf_device.h"

namespace ekf {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16, LS = 144;

// ---------------------------------------------------------------------------------------------
// k_gemm_f64: C = A B (TRANS_B = false, B row-major [k][j]) or C = A B^T + Q (TRANS_B = true, B row-major [j][k]).
// Round-2 form.  What changed against the first version (48 TFLOP/s, 70-75 % MFMA-busy):
//   * every global load is a buffer load whose resource ends at row n: rows beyond the matrix read as zero in
//     hardware, entries beyond column n are zeroed by a select -- no branch around any load, so the compiler waits
//     for exactly the slab it needs;
//   * two K slabs are in flight in registers (the loads of slab t+2 are issued before slab t is multiplied, slab t+1
//     goes registers -> LDS after it): an L2 / Infinity-Cache round trip has two slab times to land instead of one;
//   * the A / B fragments of k-step kk+1 are read from LDS while the 16 MFMAs of k-step kk run (two register sets,
//     `sched_barrier` keeps the reads in front), across the slab boundary too: the workgroup barrier (s_barrier behind
//     lgkmcnt(0) only -- `__syncthreads()` would drain the loads in flight) sits one k-step before the end of a slab.
// ---------------------------------------------------------------------------------------------
typedef unsigned int uint4v_t __attribute__((ext_vector_type(4)));
typedef unsigned int uint2v_t __attribute__((ext_vector_type(2)));

// Four 16-byte pieces starting at byte offset `off` of the resource; the whole offset goes through the VGPR (the range check of a raw
// buffer covers the VGPR and immediate offsets, not the SGPR one): anything at or beyond num_records reads as zero.
template <int STEP>                                    // bytes between the four 16-byte pieces
__device__ __forceinline__ void ldb16x4(__amdgpu_buffer_rsrc_t rs, unsigned off, double (&r)[8]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint4v_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + (unsigned)STEP * q), 0, 0);
    r[2 * q] = __builtin_bit_cast(double, uint2v_t{v.x, v.y});
    r[2 * q + 1] = __builtin_bit_cast(double, uint2v_t{v.z, v.w});
  }
}

template <bool TRANS_B>
__global__ __launch_bounds__(256, 2) void k_gemm_f64(const double* __restrict__ A,
                                                  const double* __restrict__ B,
                                                  const double* __restrict__ Q, double* __restrict__ C,
                                                  int n, int ld) {
  __shared__ __attribute__((aligned(16))) double As[2][BK][LS];
  __shared__ __attribute__((aligned(16))) double Bs[2][BK][LS];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bm = blockIdx.y * BM, bn = blockIdx.x * BN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 15, lk = lane >> 4;

  double4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

  // Staging assignment (chosen so that both the global loads and the LDS stores of a wave instruction are contiguous):
  //   "row panel" = 128 rows x 16 k (A, and B when it is multiplied transposed): lanes 0-31 of a wave take the first
  //        8 k of 32 consecutive rows, lanes 32-63 the second 8 k of the same rows -- a load instruction still covers
  //        32 rows x 128 B (whole lines), a store instruction (fixed k) 32 consecutive rows = 256 contiguous bytes
  //        (k and k + 8 of one row are 36 x 256 B apart in the k-major tile: side by side in a wave they would collide);
  //   "k panel"   = 16 k x 128 cols (B row-major): thread (k = t >> 4, c = t & 15) takes the four 16-byte pieces at
  //        columns 2 c + 32 j: 16 lanes of a load / store instruction cover 256 contiguous bytes of one k row.
  const int rp_row = (lane & 31) + 32 * wave, rp_k = (lane >> 5) * 8;
  const int kp_k = tid >> 4, kp_c = (tid & 15) * 2;
  // resources end after row n-1: a row index >= n is out of range and reads as zero
  const unsigned bytes = (unsigned)n * (unsigned)ld * 8u;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A), 0, (int)bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(B), 0, (int)bytes, 0x00020000);
  const unsigned ld8 = (unsigned)ld * 8u;
  const unsigned a_lane = (unsigned)(bm + rp_row) * ld8 + (unsigned)rp_k * 8u;                       // + k0 * 8
  const unsigned b_lane = TRANS_B ? (unsigned)(bn + rp_row) * ld8 + (unsigned)rp_k * 8u             // + k0 * 8
                                  : (unsigned)kp_k * ld8 + (unsigned)(bn + kp_c) * 8u;              // + k0 * ld8, + 256 j
  double ra[2][8], rb[2][8];                           // two slabs in flight

  auto fetch = [&](int k0, double (&xa)[8], double (&xb)[8]) {
    ldb16x4<16>(rsA, a_lane + (unsigned)k0 * 8u, xa);
    if (TRANS_B) ldb16x4<16>(rsB, b_lane + (unsigned)k0 * 8u, xb);
    else ldb16x4<256>(rsB, b_lane + (unsigned)k0 * ld8, xb);
  };
  auto stash_a = [&](int buf, int k0, const double (&xa)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) As[buf][rp_k + q][rp_row] = (k0 + rp_k + q < n) ? xa[q] : 0.0;   // columns beyond n: padding
  };
  auto stash_b = [&](int buf, int k0, const double (&xb)[8]) {
    if (TRANS_B) {
#pragma unroll
      for (int q = 0; q < 8; ++q) Bs[buf][rp_k + q][rp_row] = (k0 + rp_k + q < n) ? xb[q] : 0.0;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)                       // (rows k >= n read as zero; columns >= n are never stored)
        *reinterpret_cast<double2*>(&Bs[buf][kp_k][kp_c + 32 * j]) = make_double2(xb[2 * j], xb[2 * j + 1]);
    }
  };
  auto wg_barrier = [&]() {                            // LDS stores landed, then s_barrier: the loads of slab t+2 stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  const int nk = (n + BK - 1) / BK;
  double af[2][4], bf[2][4];                           // fragments of k-step kk in set kk & 1 (BK / 4 is even)
  fetch(0, ra[0], rb[0]);
  fetch(BK, ra[1], rb[1]);                             // (beyond the last slab: out of range or masked, harmless)
  stash_a(0, 0, ra[0]);
  stash_b(0, 0, rb[0]);
  wg_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) af[0][i] = As[0][lk][wm + i * 16 + li];
#pragma unroll
  for (int j = 0; j < 4; ++j) bf[0][j] = Bs[0][lk][wn + j * 16 + li];
  // One K slab: 4 k-steps of 16 MFMAs.  Between them, in this order: the loads of slab t+2 (register set t & 1, free
  // since slab t went to LDS), slab t+1 registers -> the other LDS buffer, the workgroup barrier one k-step before the
  // end (every fragment of slab t has been read by then), and during the last k-step the first fragments of slab t+1:
  // no MFMA of a slab waits for the barrier or for LDS.
  auto slab = [&](int t, auto par_tag) {
    constexpr int PAR = decltype(par_tag)::value;      // t & 1: LDS buffer of slab t, register set of slab t+2
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      if (kk + 1 < BK / 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[(kk + 1) & 1][i] = As[PAR][4 * (kk + 1) + lk][wm + i * 16 + li];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[(kk + 1) & 1][j] = Bs[PAR][4 * (kk + 1) + lk][wn + j * 16 + li];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[0][i] = As[1 - PAR][lk][wm + i * 16 + li];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[0][j] = Bs[1 - PAR][lk][wn + j * 16 + li];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
      if (kk == 0) fetch((t + 2) * BK, ra[PAR], rb[PAR]);
      if (kk == 1) stash_a(1 - PAR, (t + 1) * BK, ra[1 - PAR]);
      if (kk == 2) {
        stash_b(1 - PAR, (t + 1) * BK, rb[1 - PAR]);
        wg_barrier();
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int t = 0;
  for (; t + 1 < nk; t += 2) {
    slab(t, std::integral_constant<int, 0>{});
    slab(t + 1, std::integral_constant<int, 1>{});
  }
  if (t < nk) slab(t, std::integral_constant<int, 0>{});

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
