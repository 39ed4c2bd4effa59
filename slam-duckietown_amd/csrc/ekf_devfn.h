// Device-side helpers shared by the kernel translation units (ekf_kernels.hip, ekf_cadence.hip): LDS hand-off
// macros, the measurement model of src/replay_no_ros.py:443-469, buffer-instruction accessors.  gfx950 only.
#pragma once
#include "ekf_device.h"

// LDS hand-off between lanes of ONE wave (the other waves of the workgroup have exited): LDS
// operations of a wave execute in issue order, so only the compiler must be kept from reordering.
#define WAVE_SYNC()                                              \
  do {                                                           \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");       \
    __builtin_amdgcn_wave_barrier();                             \
  } while (0)

// LDS hand-off between two waves of a workgroup: the writer's LDS stores have landed (lgkmcnt(0)) before it
// flips the slot word, the reader issues its LDS loads after it saw the word; the "memory" clobber keeps the
// compiler from moving LDS accesses across.  Unlike a workgroup fence this leaves the wave's global loads
// and stores in flight (a fence drains vmcnt too).
#define LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// Workgroup barrier for data handed over through LDS only: the wave's LDS operations have completed, its global
// loads and stores stay in flight (`__syncthreads()` and workgroup fences drain vmcnt as well: a store
// acknowledgement is ~500 cycles away).  WAVE_LDS_SYNC: the same between lanes of one wave, where LDS operations
// execute in issue order and only the compiler must be kept from reordering them.
#define WG_LDS_BARRIER()                                         \
  do {                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           \
    __builtin_amdgcn_s_barrier();                                \
    asm volatile("" ::: "memory");                               \
  } while (0)
#define WAVE_LDS_SYNC()                                          \
  do {                                                           \
    asm volatile("" ::: "memory");                               \
    __builtin_amdgcn_wave_barrier();                             \
    asm volatile("" ::: "memory");                               \
  } while (0)

namespace ekf {

typedef double double4_t __attribute__((ext_vector_type(4)));

// W is stored in MFMA A-operand tiles: k-tile (4 ranks) major, then 16-row groups, then [k&3][row&15],
// so that one 8-byte-per-lane load of 64 consecutive doubles is exactly one 16x4 A fragment.
__host__ __device__ __forceinline__ long wm_index(int ld16, int k, int i) {
  return ((long)(k >> 2) * ld16 + (i >> 4)) * 64 + (k & 3) * 16 + (i & 15);
}

__device__ __forceinline__ double wrap_pi(double a) {
  // (a + pi) % (2 pi) - pi with NumPy remainder semantics, result in [-pi, pi)
  // (src/replay_no_ros.py:397, :458).  fma(-k, 2pi, x) is the exact remainder when k is the right
  // quotient (the remainder is representable); the two fix-ups cover a quotient that is off by one.
  const double two_pi = 2.0 * M_PI;
  const double x = a + M_PI;
  const double k = floor(x * (1.0 / two_pi));
  double r = fma(-k, two_pi, x);
  if (r < 0.0) r += two_pi;
  else if (r >= two_pi) r -= two_pi;
  return r - M_PI;
}

// 1 / x from the hardware estimate (v_rcp_f64: ~1e-8 relative... in fact 2^-26) and two Newton steps -- each squares the error:
// <= 1 ulp for finite NORMAL x whose reciprocal is normal too; inf / NaN / 0 behave like the IEEE division (1/0 = inf,
// 1/inf = 0, NaN through).  Where the estimate itself is not finite or zero -- x subnormal (v_rcp_f64 gives inf there), 0 or
// inf -- or x is so large that 1/x is subnormal, the Newton terms would turn the estimate into NaN / lose its bits: those take
// the IEEE division (a branch that is never taken on the solve's chain: S = H P H^T + Q with Q = meas_sigma^2 > 0).  The IEEE
// division behind `1.0 / x` is a v_div_scale / v_div_fmas / v_div_fixup sequence of ~250 dependent cycles on the solve's chain.
__device__ __forceinline__ double fast_recip(double x) {
  double r = __builtin_amdgcn_rcp(x);
  const double e0 = fma(-x, r, 1.0);
  const double r1 = fma(r, e0, r);
  const double e1 = fma(-x, r1, 1.0);
  const double r2 = fma(r1, e1, r1);
  const double ax = fabs(x);
  // (2.2250738585072014e-308 = DBL_MIN; 1 / x is normal for |x| <= 2^1022)
  if (!(ax >= 2.2250738585072014e-308 && ax <= 4.49423283715578976932e307)) return 1.0 / x;
  return r2;
}

// Innovation and 2x5 Jacobian of one range/bearing observation (src/replay_no_ros.py:443-469), in two parts: the
// Jacobian (needed first, by the covariance chain) and the innovation (atan2: twice as long, needed only by the mean).
// h[r][k] = row r, column k on {x, y, theta, lx, ly}.  q == 0 gives NaN rows like NumPy's 0/0.
struct LinGeom {
  double dx, dy, th, sq;
};
__device__ __forceinline__ LinGeom linearize_h(double mx, double my, double mth, double lx, double ly, double (&h)[2][5]) {
  LinGeom g;
  g.dx = lx - mx;                                                 // :443
  g.dy = ly - my;
  g.th = mth;
  const double dx = g.dx, dy = g.dy;
  const double q = dx * dx + dy * dy;                             // :446
  // 1/sqrt(q) once (hardware estimate + two Newton steps, <= 1 ulp); sqrt(q) = q * rs, 1/q = rs * rs.
  // q == 0 -> rs = inf -> NaN rows below, like NumPy's 0/0 at :466-469.
  double rs = __builtin_amdgcn_rsq(q);
  rs = rs * fma(-0.5 * q * rs, rs, 1.5);
  rs = rs * fma(-0.5 * q * rs, rs, 1.5);
  g.sq = q * rs;
  const double rq = rs * rs;
  const double nanv = __builtin_nan("");
  h[0][0] = -rs * dx;                                             // (-sqrt(q) dx) / q
  h[0][1] = -rs * dy;
  h[0][2] = (q > 0.0) ? 0.0 : nanv;                               // .0 / q
  h[0][3] = rs * dx;
  h[0][4] = rs * dy;
  h[1][0] = dy * rq;
  h[1][1] = -dx * rq;
  h[1][2] = (q > 0.0 && q < __builtin_inf()) ? -1.0 : nanv;       // -q / q
  h[1][3] = -dy * rq;
  h[1][4] = dx * rq;
  return g;
}
typedef unsigned int uint2v_t __attribute__((ext_vector_type(2)));
typedef unsigned int uint4v_t __attribute__((ext_vector_type(4)));
// buffer access with a 32-bit byte offset per lane (one instruction, no 64-bit address arithmetic)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rs_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ double ldb8(__amdgpu_buffer_rsrc_t rs, unsigned lane_bytes, unsigned row_bytes) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)lane_bytes, (int)row_bytes, 0));
}

// lane `src` (wave-uniform) of a double, through the scalar registers
__device__ __forceinline__ double read_lane(double v, int src) {
  const uint2v_t u = __builtin_bit_cast(uint2v_t, v);
  const unsigned lo = __builtin_amdgcn_readlane(u.x, src), hi = __builtin_amdgcn_readlane(u.y, src);
  return __builtin_bit_cast(double, uint2v_t{lo, hi});
}
__device__ __forceinline__ void innovation(const LinGeom& g, double z_range, double z_bearing, double& y0, double& y1) {
  y0 = z_range - g.sq;                                            // :455
  y1 = wrap_pi(z_bearing - (atan2(g.dy, g.dx) - g.th));           // :453-458
}

constexpr int RS_QSTRIDE = 32;          // k_flush_rs: words between the per-XCD queue heads (one cache line each)

}  // namespace ekf
