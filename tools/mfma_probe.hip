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
// hybrid: waves 0-3 of a 512-thread workgroup issue MFMAs (accumulators stay in VGPRs: no AGPR copies in the loop,
// unlike k_mfma above, whose 46-47 TFLOP/s are an artefact of those copies), waves 4-7 fp64 FMAs
__global__ __launch_bounds__(512) void k_hybrid(double* out, int reps_mfma, int reps_fma, double a0, double b0) {
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  double s = 0;
  if (threadIdx.x < 256) {
    double4_t acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    for (int r = 0; r < reps_mfma; ++r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = i;
    for (int r = 0; r < reps_fma; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fma(acc[i], a, b);
    }
    for (int i = 0; i < 8; ++i) s += acc[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
// the MFMA stream of k_flush (16 A x 4 B fragments, 4 accumulators) with operands that toggle like real data
__global__ __launch_bounds__(256, 2) void k_mfma_data(double* out, const double* in, int reps) {
  double a[16], b[16][4];
  for (int t = 0; t < 16; ++t) {
    a[t] = in[(t * 5 + 0) * 256 + threadIdx.x];
    for (int c = 0; c < 4; ++c) b[t][c] = in[(t * 5 + 1 + c) * 256 + threadIdx.x];
  }
  double4_t acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (double4_t){0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[t][i], acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
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
  for (int blocks : {256, 512}) {
    for (int ratio : {0, 4, 8, 16}) {
      const int rm = 20000, rf = rm * ratio;
      float ms = timeit([&] { hipLaunchKernelGGL(k_hybrid, dim3(blocks), dim3(512), 0, 0, out, rm, rf, 1.0, 1e-3); });
      double fm = (double)blocks * 4 * rm * 4 * 2048.0, fv = (double)blocks * 256 * (double)rf * 8 * 2.0;
      printf("hybrid blocks=%4d fma/mfma reps ratio %2d: %.3f ms  MFMA %.1f TF + VALU %.1f TF = %.1f TFLOP/s\n", blocks, ratio, ms,
             fm / ms / 1e9, fv / ms / 1e9, (fm + fv) / ms / 1e9);
    }
  }
  {
    double* in; CK(hipMalloc(&in, 8 * 256 * 80));
    static double h[256 * 80];
    unsigned long long x = 88172645463325252ull;
    for (int i = 0; i < 256 * 80; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = ((double)(x >> 11) / 9007199254740992.0 - 0.5) * 1e-3; }
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    for (int blocks : {256, 512}) {
      for (int reps : {1250, 20000}) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_mfma_data, dim3(blocks), dim3(256), 0, 0, out, in, reps); });
        printf("mfma, random operands, blocks=%4d (%d wave/SIMD) reps=%6d: %.3f ms  %.1f TFLOP/s\n", blocks, blocks / 256, reps, ms,
               (double)blocks * 4 * reps * 64 * 2048.0 / ms / 1e9);
      }
    }
  }
  for (int mult : {1, 4, 16, 32}) {                    // is the MFMA-only rate a short-burst figure?
    const int rm = 20000 * mult;
    float ms = timeit([&] { hipLaunchKernelGGL(k_hybrid, dim3(256), dim3(512), 0, 0, out, rm, 0, 1.0, 1e-3); });
    printf("mfma only (1 wave/SIMD, acc in VGPRs) reps=%7d: %.3f ms  %.1f TFLOP/s\n", rm, ms, 256.0 * 4 * rm * 4 * 2048.0 / ms / 1e9);
  }
  return 0;
}
