// Development probe: fp64 MFMA issue rate on gfx950 (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int reps, double a0, double b0) {
  double4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_fma(double* out, int reps, double a0, double b0) {
  double acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  double* out; CK(hipMalloc(&out, 8 * 256 * 4096));
  const int reps = 20000;
  for (int blocks : {256, 512, 1024}) {
    float ms = timeit([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, out, reps, 1.0, 1e-3); });
    double flops = (double)blocks * 4 * reps * 4 * 2048.0;
    printf("mfma_f64_16x16x4 x4acc  blocks=%4d: %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4GHz, %d waves/SIMD)\n", blocks, ms, flops / ms / 1e9,
           ms * 1e-3 * 2.4e9 / ((double)reps * 4 * (blocks / 256.0)), blocks / 256);
    ms = timeit([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, 0, out, reps, 1.0, 1e-3); });
    printf("mfma_f64_16x16x4 x1acc  blocks=%4d: %.3f ms  %.1f TFLOP/s\n", blocks, ms, (double)blocks * 4 * reps * 2048.0 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, reps, 1.0000001, 1e-9); });
    printf("v_fma_f64 x8 chains     blocks=%4d: %.3f ms  %.1f TFLOP/s\n", blocks, ms, (double)blocks * 256 * reps * 8 * 2.0 / ms / 1e9);
  }
  return 0;
}
