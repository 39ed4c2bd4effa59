// Development microbenchmark for the covariance pass kernel (not part of the product library).
// Variants of the streaming rank-K update P += W V on B trajectories of n x n doubles.
//   hipcc -O3 --offload-arch=gfx950 tools/pass_bench.hip -o gpurun_out/pass_bench && ./pass_bench [n] [B]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int WS = 34;

// MODE bit0: nontemporal loads, bit1: nontemporal stores, 8: copy only, 16: software-pipelined (+bits)
template <int MODE> __device__ __forceinline__ double2 ldp(const double* a) {
  double2 r;
  if (MODE & 1) { r.x = __builtin_nontemporal_load(a); r.y = __builtin_nontemporal_load(a + 1); }
  else r = *reinterpret_cast<const double2*>(a);
  return r;
}
template <int MODE> __device__ __forceinline__ void stp(double* a, double2 v) {
  if (MODE & 2) { __builtin_nontemporal_store(v.x, a); __builtin_nontemporal_store(v.y, a + 1); }
  else *reinterpret_cast<double2*>(a) = v;
}
template <int KT, int UNR, int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_pass(double* __restrict__ P, const double* __restrict__ V,
                                                     const double* __restrict__ W, int n, int ld, long pstride,
                                                     int rows_per_block) {
  const int b = blockIdx.z;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int strip = blockIdx.x * WAVES + wave;
  const int i0 = blockIdx.y * rows_per_block;
  if (strip * 128 >= n || i0 >= n) return;
  const int i1 = min(n, i0 + rows_per_block);
  const int j0 = strip * 128 + lane * 2;
  if (j0 >= n) return;
  double* Pb = P + (long)b * pstride;
  const double* Vb = V + (long)b * WS * ld;
  const double* Wb = W + (long)b * ld * WS;
  double2 v[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) v[k] = *reinterpret_cast<const double2*>(Vb + (long)k * ld + j0);
  int i = i0;
  if (MODE & 16) {
    double2 p[UNR], q[UNR];
    if (i + UNR <= i1) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) p[u] = ldp<MODE>(Pb + (long)(i + u) * ld + j0);
    }
    for (; i + UNR <= i1; i += UNR) {
      const bool more = i + 2 * UNR <= i1;
      if (more) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) q[u] = ldp<MODE>(Pb + (long)(i + UNR + u) * ld + j0);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const double* w = Wb + (long)(i + u) * WS;
#pragma unroll
        for (int k = 0; k < KT; ++k) { const double wk = w[k]; p[u].x += wk * v[k].x; p[u].y += wk * v[k].y; }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) stp<MODE>(Pb + (long)(i + u) * ld + j0, p[u]);
#pragma unroll
      for (int u = 0; u < UNR; ++u) p[u] = q[u];
    }
  } else {
    for (; i + UNR <= i1; i += UNR) {
      double2 p[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const double2* src = reinterpret_cast<const double2*>(Pb + (long)(i + u) * ld + j0);
        if (MODE & 1) { p[u].x = __builtin_nontemporal_load(&src->x); p[u].y = __builtin_nontemporal_load(&src->y); }
        else p[u] = *src;
      }
      if (MODE != 8) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const double* w = Wb + (long)(i + u) * WS;
#pragma unroll
          for (int k = 0; k < KT; ++k) { const double wk = w[k]; p[u].x += wk * v[k].x; p[u].y += wk * v[k].y; }
        }
      } else {
#pragma unroll
        for (int u = 0; u < UNR; ++u) { p[u].x += v[0].x; p[u].y += v[0].y; }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        double2* dst = reinterpret_cast<double2*>(Pb + (long)(i + u) * ld + j0);
        if (MODE & 2) { __builtin_nontemporal_store(p[u].x, &dst->x); __builtin_nontemporal_store(p[u].y, &dst->y); }
        else *dst = p[u];
      }
    }
  }
  for (; i < i1; ++i) {
    double2 p = *reinterpret_cast<const double2*>(Pb + (long)i * ld + j0);
    const double* w = Wb + (long)i * WS;
#pragma unroll
    for (int k = 0; k < KT; ++k) { const double wk = w[k]; p.x += wk * v[k].x; p.y += wk * v[k].y; }
    *reinterpret_cast<double2*>(Pb + (long)i * ld + j0) = p;
  }
}

// plain float4-style copy for the ceiling: out-of-place and in-place
__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ src, double2* __restrict__ dst, long count) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  for (; i < count; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_scale_inplace(double2* __restrict__ p, long count) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  for (; i < count; i += stride) { double2 t = p[i]; t.x *= 1.0000001; t.y *= 1.0000001; p[i] = t; }
}

template <typename F>
double time_ms(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

template <int KT, int UNR, int MODE, int WAVES>
void run(const char* name, double* P, double* V, double* W, int n, int ld, int B, int rpb) {
  dim3 grid((n + 128 * WAVES - 1) / (128 * WAVES), (n + rpb - 1) / rpb, B);
  double ms = time_ms([&] { hipLaunchKernelGGL((k_pass<KT, UNR, MODE, WAVES>), grid, dim3(WAVES * 64), 0, 0, P, V, W, n, ld, (long)ld * ld, rpb); }, 7);
  double gb = 16.0 * n * n * B / 1e9;
  printf("%-34s rpb=%4d grid=%5d  %8.3f ms  %7.1f GB/s\n", name, rpb, grid.x * grid.y * grid.z, ms, gb / (ms * 1e-3));
}

int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 4003;
  int B = argc > 2 ? atoi(argv[2]) : 32;
  int ld = (n + 15) / 16 * 16;
  size_t pe = (size_t)ld * ld * B;
  double *P, *V, *W, *P2;
  CK(hipMalloc(&P, pe * 8)); CK(hipMalloc(&V, (size_t)WS * ld * B * 8)); CK(hipMalloc(&W, (size_t)WS * ld * B * 8));
  CK(hipMemset(P, 0, pe * 8)); CK(hipMemset(V, 0, (size_t)WS * ld * B * 8)); CK(hipMemset(W, 0, (size_t)WS * ld * B * 8));
  printf("n=%d ld=%d B=%d  P=%.1f MB\n", n, ld, B, pe * 8 / 1e6);
  if (B * (size_t)ld * ld * 8 < (size_t)6e9) {
    CK(hipMalloc(&P2, pe * 8));
    double ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const double2*)P, (double2*)P2, (long)(pe / 2)); }, 7);
    printf("%-34s %8.3f ms  %7.1f GB/s\n", "copy out-of-place (16B/lane)", ms, 2.0 * pe * 8 / 1e9 / (ms * 1e-3));
    CK(hipFree(P2));
  }
  {
    double ms = time_ms([&] { hipLaunchKernelGGL(k_scale_inplace, dim3(8192), dim3(256), 0, 0, (double2*)P, (long)(pe / 2)); }, 7);
    printf("%-34s %8.3f ms  %7.1f GB/s\n", "scale in place (16B/lane)", ms, 2.0 * pe * 8 / 1e9 / (ms * 1e-3));
  }
  if (B > 4) {
    for (int rpb : {32, 128}) {
      run<18, 8, 0, 4>("plain K18 UNR8 W4", P, V, W, n, ld, B, rpb);
      run<18, 8, 1, 4>("ntL K18 UNR8 W4", P, V, W, n, ld, B, rpb);
      run<18, 8, 2, 4>("ntS K18 UNR8 W4", P, V, W, n, ld, B, rpb);
      run<18, 8, 3, 4>("ntLS K18 UNR8 W4", P, V, W, n, ld, B, rpb);
      run<18, 4, 3, 4>("ntLS K18 UNR4 W4", P, V, W, n, ld, B, rpb);
      run<18, 16, 3, 4>("ntLS K18 UNR16 W4", P, V, W, n, ld, B, rpb);
      run<18, 8, 3, 8>("ntLS K18 UNR8 W8", P, V, W, n, ld, B, rpb);
      run<18, 8, 3, 2>("ntLS K18 UNR8 W2", P, V, W, n, ld, B, rpb);
      run<18, 4, 19, 4>("ntLS pipelined UNR4 W4", P, V, W, n, ld, B, rpb);
      run<18, 8, 19, 4>("ntLS pipelined UNR8 W4", P, V, W, n, ld, B, rpb);
      run<18, 8, 8 , 4>("copy-struct plain UNR8 W4", P, V, W, n, ld, B, rpb);
      run<2, 8, 3, 4>("ntLS K2 UNR8 W4", P, V, W, n, ld, B, rpb);
      run<34, 4, 3, 4>("ntLS K34 UNR4 W4", P, V, W, n, ld, B, rpb);
    }
    for (int rpb : {16, 64, 256, 512}) run<18, 8, 3, 4>("ntLS K18 UNR8 W4", P, V, W, n, ld, B, rpb);
  } else {
    for (int rpb : {8, 16, 24, 32, 48, 64}) {
      run<18, 8, 0, 4>("plain K18 UNR8 W4", P, V, W, n, ld, B, rpb);
      run<18, 4, 0, 4>("plain K18 UNR4 W4", P, V, W, n, ld, B, rpb);
      run<18, 8, 0, 8>("plain K18 UNR8 W8", P, V, W, n, ld, B, rpb);
      run<18, 4, 0, 8>("plain K18 UNR4 W8", P, V, W, n, ld, B, rpb);
      run<18, 4, 0, 16>("plain K18 UNR4 W16", P, V, W, n, ld, B, rpb);
      run<18, 4, 16, 8>("pipelined K18 UNR4 W8", P, V, W, n, ld, B, rpb);
      run<18, 8, 2, 8>("ntS K18 UNR8 W8", P, V, W, n, ld, B, rpb);
      run<2, 8, 0, 8>("plain K2 UNR8 W8", P, V, W, n, ld, B, rpb);
    }
  }
  return 0;
}
