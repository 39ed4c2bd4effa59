// Development probe: single-wave latencies that bound k_solve / k_panels (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)

__global__ void k_probe(double* out, unsigned long long* cyc, double seed, int reps) {
  __shared__ double lds[256];
  const int lane = threadIdx.x;
  lds[lane] = seed + lane;
  lds[lane + 64] = seed * 0.5 + lane;
  __syncthreads();
  unsigned long long t[12];
  double x = seed, a = 1.0000001, b = 1e-9;
  t[0] = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) x = fma(x, a, b);                       // dependent fma chain
  asm volatile("" :: "v"(x));
  t[1] = __builtin_amdgcn_s_memtime();
  double y0 = seed, y1 = seed + 1, y2 = seed + 2, y3 = seed + 3;
  for (int i = 0; i < reps; ++i) { y0 = fma(y0, a, b); y1 = fma(y1, a, b); y2 = fma(y2, a, b); y3 = fma(y3, a, b); }
  asm volatile("" :: "v"(y0), "v"(y1), "v"(y2), "v"(y3));
  t[2] = __builtin_amdgcn_s_memtime();
  double z = seed * 0.3;
  for (int i = 0; i < reps / 16; ++i) z = atan2(z + 0.1, 1.0 + z * 0.5);  // dependent atan2
  asm volatile("" :: "v"(z));
  t[3] = __builtin_amdgcn_s_memtime();
  double w = seed + 2.0;
  for (int i = 0; i < reps / 16; ++i) w = sqrt(w + 1.0);                  // dependent sqrt
  asm volatile("" :: "v"(w));
  t[4] = __builtin_amdgcn_s_memtime();
  double r = seed + 2.0;
  for (int i = 0; i < reps / 16; ++i) r = 1.0 / (r + 0.5);                // dependent division
  asm volatile("" :: "v"(r));
  t[5] = __builtin_amdgcn_s_memtime();
  int idx = lane;
  double acc = 0;
  for (int i = 0; i < reps / 16; ++i) { double v = lds[idx & 127]; acc += v; idx = (int)v & 63; }   // dependent LDS read
  asm volatile("" :: "v"(acc));
  t[6] = __builtin_amdgcn_s_memtime();
  double s = seed * 0.1;
  for (int i = 0; i < reps / 16; ++i) { double sn, cs; sincos(s, &sn, &cs); s = sn * 0.5 + cs * 0.1; }
  asm volatile("" :: "v"(s));
  t[7] = __builtin_amdgcn_s_memtime();
  unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  double q = seed;
  for (int i = 0; i < reps * 4; ++i) q = fma(q, a, b);
  asm volatile("" :: "v"(q));
  unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
  t[8] = __builtin_amdgcn_s_memtime();
  if (lane == 0) {
    for (int i = 0; i < 9; ++i) cyc[i] = t[i];
    cyc[9] = rt1 - rt0;
    out[0] = x + y0 + y1 + y2 + y3 + z + w + r + acc + s + q;
  }
}

int main() {
  double* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 128));
  const int reps = 4096;
  for (int it = 0; it < 3; ++it) {
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, out, cyc, 0.37, reps);
    CK(hipDeviceSynchronize());
  }
  unsigned long long h[16];
  CK(hipMemcpy(h, cyc, 80, hipMemcpyDeviceToHost));
  printf("dependent fma_f64      : %.1f cycles each\n", double(h[1] - h[0]) / reps);
  printf("4 independent fma_f64  : %.1f cycles per fma\n", double(h[2] - h[1]) / (4.0 * reps));
  printf("dependent atan2        : %.1f cycles each\n", double(h[3] - h[2]) / (reps / 16));
  printf("dependent sqrt(+add)   : %.1f cycles each\n", double(h[4] - h[3]) / (reps / 16));
  printf("dependent 1/x (+add)   : %.1f cycles each\n", double(h[5] - h[4]) / (reps / 16));
  printf("dependent LDS read     : %.1f cycles each\n", double(h[6] - h[5]) / (reps / 16));
  printf("dependent sincos       : %.1f cycles each\n", double(h[7] - h[6]) / (reps / 16));
  double cycles = double(h[8] - h[7]);
  double ns = double(h[9]) * 10.0;   // s_memrealtime ticks at 100 MHz
  printf("clock during single-wave fma loop: %.0f MHz (%.0f cycles in %.0f ns)\n", cycles / ns * 1e3, cycles, ns);
  return 0;
}
