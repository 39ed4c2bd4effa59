// Shared host/device records of the EKF-SLAM core (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ekfslam_hip.h"

namespace ekf {

constexpr int MMAX = EKF_MMAX;          // landmarks per update pass
constexpr int CMAX = 3 + 2 * MMAX;      // compressed sub-state size (35)
constexpr int TS = 36;                  // row stride of T (>= CMAX)
constexpr int US = 2 * MMAX;            // row stride of U (32)

constexpr int FLAG_PREDICT = 1;         // StepIn.flags
constexpr int FLAG_UPDATE = 2;

// One step's inputs for one trajectory (host -> device, 344 B).
struct StepIn {
  double lin, ang;
  int m;
  int flags;
  int idx[MMAX];
  double range[MMAX];
  double bearing[MMAX];
};

// Output of the sequential compressed solve for one trajectory; read by the panel and pass kernels.
struct SolveOut {
  double g[2];          // G[0,2], G[1,2] of the motion Jacobian (0 when prediction is off)
  double rd[3];         // motion noise added to the pose block (0 when prediction is off)
  double p22h;          // 0.5 * P[2,2] before the step
  int c;                // 3 + 2m
  int m;
  int C[CMAX + 1];      // gathered state indices, padded with 0
  double mu_c[CMAX];    // updated mean entries at C
  double ys[US];        // stacked innovations
  double T[US][TS];     // V = T P'[C,:]
  double U[CMAX][US];   // Kst = P'[:,C] U
};

struct DeviceConfig {
  double rd[3];         // diag of R  (src/replay_no_ros.py:421)
  double qd[2];         // diag of Q  (:438)
  double arc_threshold; // :376
  int enable_measurement_model, enable_circular_interpolation, disable_motion_model;
};

}  // namespace ekf
