// Shared host/device records of the EKF-SLAM core (gfx950 only).
#pragma once
#ifdef EKF_HOST_ONLY                                     /* plain g++ (the sanitizer build of the host logic, tests/host_plan_check.cpp) */
#define __host__
#define __device__
#define __forceinline__ inline
#else
#include <hip/hip_runtime.h>
#endif
#include <cstddef>
#include "../../include/ekfslam_hip.h"

namespace ekf {

constexpr int MMAX = EKF_MMAX;          // landmarks per update pass
constexpr int CMAX = 3 + 2 * MMAX;      // compressed sub-state size (35)
constexpr int KTOT = 80;                // rank slots of the deferred low-rank update  P = P_base + W V
constexpr int NKT = KTOT / 4;           // the same in MFMA k-tiles (v_mfma_f64_16x16x4)

// ranks one step appends: 2 per observed landmark (the prediction only changes rows 0,1 of the stored
// triangle and is applied to P_base directly); steps are packed back to back and only the total is padded
// to a whole k-tile (the step that is last so far zero-fills the pad ranks)
__host__ __device__ constexpr int ranks_for(int mcap) { return 2 * mcap; }

// ---- layout of one trajectory's P_base (round 4) ----
// Row-major with row stride ld while ld <= 4096 (ld is then a power of two).  Beyond that the covariance is cut into
// COLUMN PANELS of PPW = 4096 doubles: panel p holds the columns [4096 p, 4096 (p + 1)) of every row, row-major with a row
// stride of exactly 4096 doubles (32 KB), panels ld x 4096 doubles apart (rows == ld there):
//     index(i, j) = (j >> 12) * (ld * 4096) + i * min(ld, 4096) + (j & 4095)
// -- one formula for both cases (j < ld <= 4096 leaves the panel term at zero).  Why: the sixteen 512-byte row segments of
// a 16 x 64 tile of the covariance pass then lie 32 KB apart whatever the size of the state, the pitch the pass streams
// best at (N = 2000, ld = 4096: 6.05 TB/s; with rows 128 KB apart at N = 8000, ld = 16064: 5.0 - 5.4 TB/s).  A 64-column
// strip never straddles a panel (4096 is a multiple of 64), so a kernel that walks a row strip by strip only needs the
// strip's column offset p_col(ld, j0) and the row stride p_lds(ld).  V, W and the mean keep the plain stride ld.
constexpr int PPW = 4096;
#ifdef P_ROWMAJOR_PROBE                                  /* diagnostic build: the plain row-major layout at every size (A/B timing) */
__host__ __device__ __forceinline__ int p_lds(int ld) { return ld; }
__host__ __device__ __forceinline__ long p_col(int ld, int j) { return j; }
__host__ __device__ __forceinline__ unsigned p_col8(int ld, unsigned j) { return j * 8u; }
#else
__host__ __device__ __forceinline__ int p_lds(int ld) { return ld < PPW ? ld : PPW; }
__host__ __device__ __forceinline__ long p_col(int ld, int j) { return (long)(j >> 12) * ((long)ld * PPW) + (j & (PPW - 1)); }
// the same as a 32-bit byte offset (ekf_create bounds one covariance by 4 GiB)
__host__ __device__ __forceinline__ unsigned p_col8(int ld, unsigned j) { return (j >> 12) * ((unsigned)ld * (unsigned)(PPW * 8)) + (j & (unsigned)(PPW - 1)) * 8u; }
#endif
__host__ __device__ __forceinline__ long p_index(int ld, int i, int j) { return p_col(ld, j) + (long)i * p_lds(ld); }
// panels / doubles allocated per trajectory (every panel keeps all `rows` rows: downloads and the dense product get the
// mirrored matrix in place)
__host__ __device__ __forceinline__ int p_panels(int ld) { return ld <= PPW ? 1 : (ld + PPW - 1) / PPW; }
__host__ __device__ __forceinline__ long p_alloc(int rows, int ld) { return ld <= PPW ? (long)rows * ld : (long)p_panels(ld) * rows * PPW; }

constexpr int FLAG_PREDICT = 1;         // StepIn.flags
constexpr int FLAG_UPDATE = 2;

// One step's inputs for one trajectory (host -> device, 352 B).
struct StepIn {
  double lin, ang;
  int m;
  int flags;
  int neff;             // active bound: state indices >= neff have never been correlated with anything
  int pad;
  int idx[MMAX];
  double range[MMAX];
  double bearing[MMAX];
};

// What one sequential iteration (one observed landmark j, src/replay_no_ros.py:436-480) hands to
// the panel kernel, expressed on the compressed index set C.  Pairs are stored adjacent so that a
// single 16-byte LDS broadcast read fetches both rows of a 2-row quantity.
struct alignas(16) SolveIter {
  double h5t[5][2];     // {H[0][k], H[1][k]}: the 2x5 Jacobian on {0,1,2,t_j,t_j+1}    (:466-469)
  double si[4];         // S_j^{-1} row-major                                             (:473)
  double y[2];          // innovation                                                     (:455-458)
  double kc[CMAX][2];   // K_j[C[a],:]                 (0 for a >= c)
};
static_assert(sizeof(SolveIter) % 16 == 0, "SolveIter must stay 16-byte granular");

// The pending factors restricted to C, written by the solve kernel for the panel kernel's scalar loads:
// per trajectory  facW[a][k] = W[C[a]][k]  then  facV[a][k] = V[k][C[a]],  both [CMAX][KTOT], zero-filled
// up to the next multiple of 8 ranks.
constexpr int FACS = 2 * CMAX * KTOT;

// Output of the sequential compressed solve for one trajectory; read by the panel and pass kernels.
struct alignas(16) SolveHead {
  double g[2];          // G[0,2], G[1,2] of the motion Jacobian (0 when prediction is off)
  double rd[3];         // motion noise added to the pose block (0 when prediction is off)
  double p22h;          // 0.5 * P[2,2] before the step
  double dacc_old[3];   // pose-block noise already pending before this step
  int cmax;             // largest gathered state index (panel waves starting beyond it never read W)
  int pad0;
  int c;                // 3 + 2m
  int m;
  int kbase;            // ranks pending before this step (multiple of 4)
  int neff;             // active bound of this step (rows/cols >= neff of P are untouched diagonal)
  int C[CMAX + 1];      // gathered state indices, padded with 0
  double prow[2][CMAX + 1];   // P(0, C[a]) and P(1, C[a]) before the step (what state indices 0,1 gather)
};
struct alignas(16) SolveOut : SolveHead {
  SolveIter it[MMAX];
#ifdef EKF_STAMPS
  unsigned long long stamps[128];   // diagnostic build only (tools/solve_probe.hip)
#endif
};
static_assert(sizeof(SolveHead) % 16 == 0, "SolveHead must stay 16-byte granular");
// (SolveOut = head, then the records: `it` starts at sizeof(SolveHead))

// ---- fused cadence (ekf_cadence.hip): everything between two covariance passes of an uploaded stream as ONE solve launch
// and ONE panel launch.  Right after a pass nothing is pending and the stream knows the next landmark indices: the union
// panel P(C_u, i), C_u = {0,1,2} + the landmark indices of all updates up to the next pass, is gathered from P_base once, the
// predictions and the sequential landmark updates of src/replay_no_ros.py:368-480 are replayed on it, and all ranks are
// appended at once.
// Round 5: the cadence is PACKED.  A trajectory's stream is the flat sequence  P_t, L_t,0 .. L_t,m_t-1, P_t+1, ...  (P = the
// prediction of step t, L = one landmark update); a cadence takes the next <= CAD_SLOTS landmark updates of that sequence,
// whatever steps they belong to (a step may be cut: its remaining landmarks open the next cadence, without a second
// prediction), plus every prediction in between -- exactly 2 ranks per landmark update, no rounding of a step's landmark
// count to a slot size, steps that observe nothing ride along for free.  Every trajectory of a bank walks its OWN sequence
// (trajectories are closed systems, src/replay_no_ros.py:269-482 couples nothing): after a pass each one has used its 40
// slots, however its landmark counts wander (CadPlan, planned on the host: ekf_host_plan.h).
// Slots are RIGHT-ALIGNED: a cadence of `nslots` updates uses the slots s0 = CAD_SLOTS - nslots .. CAD_SLOTS - 1; slot s sits
// at positions pa(s), pa(s) + 1 of C_u with pa(s) = 3 + 2 (CAD_SLOTS - 1 - s): later slots at LOWER positions, the last one
// always at 3, 4 -- "everything a later update still needs" is always the prefix [0, pa(s)) of the positions: compile-time
// bounds for the panel's register array, a shrinking prefix of lanes for the solve, and a short cadence costs what its
// slots cost.  A landmark observed twice has two slots (duplicate positions carry identical values); positions beyond
// 3 + 2 nslots gather index 0 and are never read.
constexpr int CAD_SLOTS = KTOT / 2;                  // landmark slots of a cadence (2 ranks each)
constexpr int CAD_CU = 3 + 2 * CAD_SLOTS;            // gathered positions at most (83)
__host__ __device__ constexpr int cad_pa(int s) { return 3 + 2 * (CAD_SLOTS - 1 - s); }
// record of slot s (doubles): h5t[5][2], si[4], y[2], then K_s[C_u[a], :] for the positions a < pa(s) a later slot or
// step still reads; records are packed back to back
__host__ __device__ constexpr int cad_rec_off(int s) { return 16 * s + 2 * (s * (3 + 2 * (CAD_SLOTS - 1)) - s * (s - 1)); }
struct CadGeom {
  static constexpr int GM = CAD_SLOTS;               // landmark slots
  static constexpr int CU = CAD_CU;                  // gathered positions
  __host__ __device__ static constexpr int pa(int s) { return cad_pa(s); }
  __host__ __device__ static constexpr int rec_off(int s) { return cad_rec_off(s); }
  static constexpr int REC = cad_rec_off(CAD_SLOTS);
};
constexpr int CAD_REC_MAX = CadGeom::REC;            // 4000 doubles
// What one trajectory does in one cadence (host -> device, 32 B): the steps t0 .. t0 + ns - 1 of the uploaded stream; of the
// first one the landmarks from j0 on (j0 > 0: the step was cut by the previous cadence, its prediction has happened), of
// the last one the landmarks up to jend (exclusive; the rest open the next cadence).  ns == 0: nothing (the trajectory has
// reached the end of the range; its ranks of this cadence are zero).
struct CadPlan {
  int t0, j0, ns, jend;
  int nslots;                 // landmark updates (<= CAD_SLOTS)
  int neff;                   // active bound of the cadence (its last step's, raised to the handle's floor)
  int pad[2];
};
struct alignas(16) CadHead {
  int nslots;                 // landmark slots in use: CAD_SLOTS - nslots .. CAD_SLOTS - 1
  int neff;                   // active bound of the cadence
  int npred;                  // steps touched (a prediction each, except a first step cut by the previous cadence)
  int pad0;
  int sfirst[CAD_SLOTS];      // slot of the first landmark of touched step p, or of the next one that has any (CAD_SLOTS: none)
  int C[CAD_CU + 1];          // gathered state indices by position
  double g[CAD_SLOTS][2];     // G[0,2], G[1,2] of touched step p's motion Jacobian (0 when it predicts nothing)
  double prow[2][CAD_CU + 1]; // (diagnostic) P(0, C_u[a]), P(1, C_u[a]) before the cadence
  double ddpose[2][4];        // what the cadence's predictions add to P_base(0, l), P_base(1, l), l < 3 (the pose block)
  double rdsum[4];            // pose-block noise of the cadence's predictions (applied by the panel launch when no rank is pending)
};
struct alignas(16) CadOut : CadHead {
  double rec[CAD_REC_MAX];
  double posevw[CAD_SLOTS][3][4];   // the new ranks' entries at the pose's state indices l < 3: V[2s][l], V[2s+1][l], W[l][2s], W[l][2s+1]
  double posefin[4][4];             // (k_solve_cad<true>: chained runs) the pose block P(l, l') after the cadence, motion noise included
};
// (chained runs) a cadence's inputs as k_solve_cad needs them -- positions, per-step counts and slots, motion inputs,
// measurements -- formed ONE CADENCE AHEAD by an otherwise idle workgroup of the chain launch (cad_positions is two dependent
// memory round trips: 2.5 us at the head of the solve, of the chain launch and of its gather workgroups)
struct alignas(16) CadPre {
  int nslots, pad[3];
  int C[128];                                // gathered state indices by position (0 beyond the cadence's)
  int cnt[CAD_SLOTS + 4], first[CAD_SLOTS + 4], lo[CAD_SLOTS + 4];   // per touched step: landmarks, slots in front of it, first landmark
  int fl[CAD_SLOTS];                         // per touched step: StepIn.flags (a first step cut by the previous cadence: no prediction)
  double la[CAD_SLOTS][2];                   // per touched step: (lin, ang)
  double z[CAD_SLOTS][2];                    // per slot: (range, bearing)
};
static_assert(sizeof(CadPre) % 16 == 0, "CadPre must stay 16-byte granular");
static_assert(sizeof(CadHead) % 16 == 0, "CadHead must stay 16-byte granular");
static_assert(sizeof(CadPlan) == 32, "CadPlan is 32 bytes");

// Device-side association: one window of raw AprilTag detections per trajectory.
constexpr int DMAX = EKF_DMAX;          // detections per window (256)
constexpr int AMAX = EKF_AMAX;          // distinct tags per window (32: two update passes of MMAX landmarks)
constexpr int TAGMAX = 1024;            // tag ids [0, TAGMAX) (tag36h11 has 587)
constexpr int IGNMAX = 16;

struct DetIn {                          // host -> device
  double lin, ang;
  int count;
  int pad;
  int tag_id[DMAX];
  double pose_err[DMAX];
  double pose_t[DMAX][3];
};

struct AssocOut {                       // what the reference returns as tags_positions (:331-337), update order
  int m;
  int n_after;
  int idx[AMAX];
  int tag_id[AMAX];
  double xw[AMAX], yw[AMAX], err[AMAX], range[AMAX], bearing[AMAX];
};

struct AssocConfig {
  double gate2;                         // 1.5^2 (:289)
  double init_var;                      // 1e4   (:356-357)
  int n_ignore;
  int active_bound;
  int ignore[IGNMAX];                   // IGNORE_TAGS (:36, :286)
};

struct DeviceConfig {
  double rd[3];         // diag of R  (src/replay_no_ros.py:421)
  double qd[2];         // diag of Q  (:438)
  double arc_threshold; // :376
  int enable_measurement_model, enable_circular_interpolation, disable_motion_model;
};

}  // namespace ekf
