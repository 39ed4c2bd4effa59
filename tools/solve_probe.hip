// Diagnostic: where k_solve spends its cycles (s_memtime stamps).  Not part of the product.
//   hipcc -O3 --offload-arch=gfx950 -DEKF_STAMPS tools/solve_probe.hip -o tools/solve_probe
#include "../slam-duckietown_amd/csrc/ekf_kernels.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace ekf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
int main(int argc, char** argv) {
  const int N = 2000, n = 3 + 2 * N, ld = (n + 15) / 16 * 16, m = argc > 1 ? atoi(argv[1]) : 8, kbase = argc > 2 ? atoi(argv[2]) : 0;
  double *P, *mu0, *mu1, *V, *W, *dacc, *fac; int* dn; StepIn* din; SolveOut* dso; unsigned* dfl;
  CK(hipMalloc(&P, sizeof(double) * ld * ld)); CK(hipMalloc(&mu0, sizeof(double) * ld)); CK(hipMalloc(&mu1, sizeof(double) * ld));
  CK(hipMalloc(&V, sizeof(double) * KTOT * ld)); CK(hipMalloc(&W, sizeof(double) * KTOT * ld)); CK(hipMemset(V, 0, sizeof(double) * KTOT * ld)); CK(hipMemset(W, 0, sizeof(double) * KTOT * ld)); CK(hipMalloc(&dacc, 64)); CK(hipMemset(dacc, 0, 64));
  CK(hipMalloc(&fac, sizeof(double) * FACS)); CK(hipMalloc(&dn, 4)); CK(hipMalloc(&din, sizeof(StepIn))); CK(hipMalloc(&dso, sizeof(SolveOut))); CK(hipMalloc(&dfl, 4)); unsigned* dq; CK(hipMalloc(&dq, 4 * 8 * RS_QSTRIDE));
  std::vector<double> hP((size_t)ld * ld, 0.0), hmu(ld, 0.0);
  for (int i = 0; i < n; ++i) { hP[(size_t)i * ld + i] = i < 3 ? 0.1 : 1e4; if (i >= 3) hmu[i] = 0.3 + 0.001 * i * ((i & 1) ? 1 : -1); }
  CK(hipMemcpy(P, hP.data(), sizeof(double) * ld * ld, hipMemcpyHostToDevice));
  CK(hipMemcpy(mu0, hmu.data(), sizeof(double) * ld, hipMemcpyHostToDevice));
  CK(hipMemcpy(dn, &n, 4, hipMemcpyHostToDevice)); CK(hipMemset(dfl, 0, 4));
  StepIn s{}; s.lin = 0.004; s.ang = 0.02; s.m = m; s.flags = 3; s.neff = n;
  for (int i = 0; i < m; ++i) { s.idx[i] = 7 * i + 3; s.range[i] = 0.8; s.bearing[i] = 0.1 * i; }
  CK(hipMemcpy(din, &s, sizeof(s), hipMemcpyHostToDevice));
  DeviceConfig cfg{}; cfg.rd[0] = cfg.rd[1] = 0.01; cfg.rd[2] = 0.0025; cfg.qd[0] = cfg.qd[1] = 0.49; cfg.arc_threshold = 1e-2;
  cfg.enable_measurement_model = 1; cfg.enable_circular_interpolation = 1; cfg.disable_motion_model = 0;
  SolveOut ho;
  for (int rep = 0; rep < 3; ++rep) {
    launch_solve(0, P, V, W, dacc, dacc + 4, mu0, mu1, dn, din, dso, dfl, fac, dn, dq, cfg, ld, (long)ld * ld, 1, kbase);   // (floor := n)
    CK(hipDeviceSynchronize());
  }
  CK(hipMemcpy(&ho, dso, sizeof(ho), hipMemcpyDeviceToHost));
  auto d = [&](int a, int b) { return (long long)(ho.stamps[b] - ho.stamps[a]); };
  printf("m=%d  pending ranks %d  total %lld cycles\n", m, kbase, d(0, 6));
  printf("inputs+sync %lld | gather %lld | motion %lld | predict+writeLDS %lld | first linearize %lld\n", d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5));
  printf("   inside gather: base loads issued %lld | factor loads issued %lld | motion model %lld | factors staged+barrier %lld | MFMA+barrier %lld | rest %lld\n",
         d(1, 110), 0LL, d(110, 112), d(112, 113), kbase > 0 ? d(113, 114) : 0LL, kbase > 0 ? d(114, 2) : d(113, 2));
  for (int j = 0; j < m; ++j) {
    int b0 = 8 + 6 * j;
    long long prev = j == 0 ? (long long)ho.stamps[5] : (long long)ho.stamps[12 + 6 * (j - 1)];
    printf("iter %2d: gap %5lld | A %5lld | B %5lld | linearize %5lld | downdate %5lld\n", j, (long long)ho.stamps[b0] - prev, d(b0, b0 + 1), d(b0 + 1, b0 + 2), d(b0 + 2, b0 + 3), d(b0 + 3, b0 + 4));
  }
  printf("tail %lld\n", d(12 + 6 * (m - 1), 6));
  return 0;
}
