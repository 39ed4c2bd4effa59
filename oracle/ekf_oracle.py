"""CPU oracle for the EKF-SLAM predict/update hot path.  TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the reference algorithm.  It is the checker
that the HIP path is compared against; it is never the thing that is shipped or
measured as the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.

Parity pin: the reference has no tests of its own.  The
pin is ``tests/golden/*.npz``: vectors produced in the build container by
executing the reference's own ``EKF_pose_estimation`` / ``predict`` / ``update``
function bodies (``oracle/gen_golden.py``), and ``tests/test_oracle_golden.py``
checks every function below against them.

Two restatements live here:

* ``*_dense``      reference-shaped: the same dense products, in the same order,
                   as ``src/replay_no_ros.py:420-480``.  This is the CPU
                   baseline that ``bench.py`` times ("kind": "port").
* ``*_structured`` the O(n^2) formulation the HIP kernels implement (compressed
                   c x c sequential solve, V = T P'[C,:], W = -P'[:,C] U, one rank-K
                   pass).  It
                   is itself pinned to ``*_dense`` and to the golden vectors, and
                   is what the GPU is compared against at sizes where the dense
                   path would take minutes.

All citations are relative to /root/reference/.
"""
from __future__ import annotations

import dataclasses
from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import numpy as np

TWO_PI = 2.0 * np.pi


@dataclasses.dataclass
class EkfConfig:
    """Module constants of src/replay_no_ros.py:15-36 as explicit fields."""

    motion_sigma: float = 0.1            # MOTION_MODEL_VARIANCE       :15
    meas_sigma: float = 0.7              # MEASUREMENT_MODEL_VARIANCE  :16
    enable_measurement_model: bool = True      # :18
    enable_circular_interpolation: bool = True  # :19
    disable_motion_model: bool = False         # :28
    arc_threshold: float = 1e-2          # literal at :376
    gate_range: float = 1.5              # literal at :289
    landmark_init_var: float = 10000.0   # literal at :356-357
    ignore_tags: Tuple[int, ...] = ()    # IGNORE_TAGS :36

    def motion_noise_diag(self) -> np.ndarray:
        # :421  R = diag(s^2, s^2, (s/2)^2)
        s = self.motion_sigma
        return np.array([s ** 2, s ** 2, (s / 2) ** 2])

    def meas_noise_diag(self) -> np.ndarray:
        # :438  Q = diag(s^2, s^2)
        s = self.meas_sigma
        return np.array([s ** 2, s ** 2])


def wrap_pi(a):
    """(a + pi) % 2pi - pi with NumPy remainder semantics -> [-pi, pi).  :397, :458"""
    return (a + np.pi) % TWO_PI - np.pi


# --------------------------------------------------------------------------
# a11: odometry scalars  (src/replay_no_ros.py:250-266, :484-497)
# --------------------------------------------------------------------------
def delta_phi(ticks: int, prev_ticks: int, resolution: int) -> float:
    return (ticks - prev_ticks) * (TWO_PI / resolution)


def displacement(wheel_radius: float, baseline: float, dphi_left: float, dphi_right: float):
    d_r = wheel_radius * dphi_right
    d_l = wheel_radius * dphi_left
    return (d_r - d_l) / baseline, (d_l + d_r) / 2      # (angular, linear)


# --------------------------------------------------------------------------
# a2: association, gate, averaging  (src/replay_no_ros.py:280-337)
# --------------------------------------------------------------------------
def associate(detections, tag_index: Dict[int, int], pose: np.ndarray, cfg: EkfConfig):
    """detections: [(timestamp, [tag...])], tag has .tag_id .pose_R .pose_t(3,1) .pose_err.

    Mutates tag_index like the reference (:294-295).  Returns an insertion-ordered
    dict  landmark_idx -> [xw, yw, err, tag_id, range, bearing]  (:331-337); the
    world guess uses the pose BEFORE prediction.
    """
    buckets: "OrderedDict[int, list]" = OrderedDict()
    for _ts, tags in detections:
        for tag in tags:
            if tag.tag_id in cfg.ignore_tags:                      # :286
                continue
            if tag.pose_t[2][0] ** 2 + tag.pose_t[0][0] ** 2 > cfg.gate_range ** 2:   # :289
                continue
            if tag.tag_id not in tag_index:                        # :294-295
                tag_index[tag.tag_id] = len(tag_index)
            buckets.setdefault(tag_index[tag.tag_id], []).append((tag.pose_t, tag.pose_err))
    inverse = {v: k for k, v in tag_index.items()}                 # :311
    out: "OrderedDict[int, list]" = OrderedDict()
    for idx, obs in buckets.items():
        t = np.mean([o[0] for o in obs], axis=0)                   # :315
        err = np.mean([o[1] for o in obs], axis=0)                 # :317
        x_r, y_r = t[2][0], -t[0][0]                               # :321
        rng = np.sqrt(x_r ** 2 + y_r ** 2)                         # :329
        brg = np.arctan2(y_r, x_r)                                 # :330
        out[idx] = [pose[0] + rng * np.cos(brg + pose[2]),         # :331
                    pose[1] + rng * np.sin(brg + pose[2]),         # :332
                    err, inverse[idx], rng, brg]
    return out


# --------------------------------------------------------------------------
# a3: augmentation  (src/replay_no_ros.py:341-360)
# --------------------------------------------------------------------------
def augment(mean: np.ndarray, cov: np.ndarray, n_landmarks: int, tags_positions, cfg: EkfConfig):
    n_new = 3 + 2 * n_landmarks
    n_old = len(mean)
    if n_old >= n_new:
        return mean, cov
    mean2 = np.zeros(n_new)
    mean2[:n_old] = mean
    cov2 = np.zeros((n_new, n_new))
    cov2[:n_old, :n_old] = cov
    for i in range(n_old, n_new, 2):
        cov2[i, i] = cfg.landmark_init_var
        cov2[i + 1, i + 1] = cfg.landmark_init_var
        j = (i - 3) // 2
        mean2[i] = tags_positions[j][0]          # KeyError if unseen, like :359
        mean2[i + 1] = tags_positions[j][1]
    return mean2, cov2


# --------------------------------------------------------------------------
# a4: motion model mean + 3x3 Jacobian  (src/replay_no_ros.py:368-417)
# --------------------------------------------------------------------------
def motion_model(pose: np.ndarray, lin: float, ang: float, cfg: EkfConfig):
    """Returns (new_pose(3,), G(3,3)).  Theta is read before the update (:370)."""
    th = pose[2]
    G = np.eye(3)
    new = np.array(pose[:3], dtype=float)
    if cfg.disable_motion_model:                                   # :372-373
        return new, G
    if cfg.enable_circular_interpolation:
        if abs(ang) <= cfg.arc_threshold:                          # :376  straight, theta NOT advanced
            new = new + np.array([lin * np.cos(th), lin * np.sin(th), 0])
            G[0, 2] = -lin * np.sin(th)
            G[1, 2] = lin * np.cos(th)
        else:                                                      # :390  arc
            r = lin / ang
            new = new + np.array([-r * np.sin(th) + r * np.sin(th + ang),
                                  r * np.cos(th) - r * np.cos(th + ang),
                                  ang])
            new[2] = wrap_pi(new[2])                               # :397
            G[0, 2] = -r * np.cos(th) + r * np.cos(th + ang)       # :401
            G[1, 2] = -r * np.sin(th) + r * np.sin(th + ang)       # :402
    else:                                                          # :405-417  no wrap
        new = new + np.array([lin * np.cos(th), lin * np.sin(th), ang])
        G[0, 2] = -lin * np.sin(th)
        G[1, 2] = lin * np.cos(th)
    return new, G


# --------------------------------------------------------------------------
# a6/a7: innovation and the 2x5 measurement Jacobian  (src/replay_no_ros.py:440-469)
# --------------------------------------------------------------------------
def innovation_and_h5(pose3: np.ndarray, lm2: np.ndarray, z_range: float, z_bearing: float):
    d = lm2 - pose3[0:2]                                           # :443
    q = d @ d                                                      # :446
    sq = np.sqrt(q)
    zhat = np.array([sq, np.arctan2(d[1], d[0]) - pose3[2]])       # :451-454
    y = np.array([z_range, z_bearing]) - zhat                      # :455
    y[1] = wrap_pi(y[1])                                           # :458
    with np.errstate(divide="ignore", invalid="ignore"):           # q == 0 -> NaN/inf like :466-469
        h5 = np.array([[-sq * d[0], -sq * d[1], 0.0, sq * d[0], sq * d[1]],
                       [d[1], -d[0], -q, -d[1], d[0]]], dtype=float) / q
    return y, h5


# --------------------------------------------------------------------------
# a5, a8, a9  reference-shaped DENSE path  (src/replay_no_ros.py:420-480)
# --------------------------------------------------------------------------
def predict_dense(mean: np.ndarray, cov: np.ndarray, lin: float, ang: float, cfg: EkfConfig):
    n = len(mean)
    mean = np.array(mean, dtype=float)
    pose, G = motion_model(mean[0:3], lin, ang, cfg)
    if not cfg.disable_motion_model:
        mean[0:3] = pose
    F = np.zeros((3, n))
    F[:, 0:3] = np.eye(3)
    GF = np.eye(n)
    GF[0:3, 0:3] = G                                               # :428-429
    cov = GF @ cov @ GF.T + F.T @ np.diag(cfg.motion_noise_diag()) @ F   # :430
    return mean, cov


def update_dense(mean: np.ndarray, cov: np.ndarray, idx: Sequence[int], ranges: Sequence[float],
                 bearings: Sequence[float], cfg: EkfConfig):
    """Sequential per-landmark update, simple-form covariance (:436-480)."""
    n = len(mean)
    mean = np.array(mean, dtype=float)
    Q = np.diag(cfg.meas_noise_diag())
    for j, zr, zb in zip(idx, ranges, bearings):
        t = 3 + 2 * int(j)                                         # :440
        y, h5 = innovation_and_h5(mean[0:3], mean[t:t + 2], zr, zb)
        H = np.zeros((2, n))
        H[:, 0:3] = h5[:, 0:3]
        H[:, t:t + 2] = h5[:, 3:5]                                 # h5 @ Fx_j, :461-469
        K = cov @ H.T @ np.linalg.inv(H @ cov @ H.T + Q)           # :473
        mean = mean + K @ y                                        # :476
        cov = (np.eye(n) - K @ H) @ cov                            # :480
    return mean, cov


def ekf_step_dense(mean, cov, lin, ang, idx, ranges, bearings, cfg: EkfConfig):
    """predict + update on a pre-sized state (what the GPU `step` is compared with)."""
    mean, cov = predict_dense(mean, cov, lin, ang, cfg)
    if cfg.enable_measurement_model:                               # :435
        mean, cov = update_dense(mean, cov, idx, ranges, bearings, cfg)
    return mean, cov


def ekf_pose_estimation_dense(ang, lin, mean, cov, delta_t, detections, tag_index, cfg: EkfConfig = None):
    """Whole-step restatement of EKF_pose_estimation (src/replay_no_ros.py:269-482).

    delta_t is accepted and unused, like the reference (:274).
    """
    cfg = cfg or EkfConfig()
    tags_positions = associate(detections, tag_index, mean, cfg)
    mean, cov = augment(np.asarray(mean, dtype=float), np.asarray(cov, dtype=float),
                        len(tag_index), tags_positions, cfg)
    idx = list(tags_positions.keys())
    rng = [tags_positions[k][4] for k in idx]
    brg = [tags_positions[k][5] for k in idx]
    mean, cov = ekf_step_dense(mean, cov, lin, ang, idx, rng, brg, cfg)
    return mean, cov, dict(tags_positions)


# --------------------------------------------------------------------------
# a10: the 3-state prototype  (src/EKF-SLAM.py:8-13, :29-84)
# --------------------------------------------------------------------------
MOTION_NOISE_3 = np.diag([0.1, 0.1, np.radians(5)])    # EKF-SLAM.py:12
OBSERVATION_NOISE_3 = np.diag([0.5, 0.5])              # EKF-SLAM.py:13


def predict3(state, covariance, control, dt):
    v, om = control
    th = state[2]
    if abs(om) > 1e-6:                                             # EKF-SLAM.py:34
        step = np.array([-v / om * np.sin(th) + v / om * np.sin(th + om * dt),
                         v / om * np.cos(th) - v / om * np.cos(th + om * dt),
                         om * dt])
    else:
        step = np.array([v * np.cos(th) * dt, v * np.sin(th) * dt, 0])
    state = state + step
    state[2] = np.arctan2(np.sin(state[2]), np.cos(state[2]))      # :45
    F = np.array([[1, 0, -v * dt * np.sin(th)],
                  [0, 1, v * dt * np.cos(th)],
                  [0, 0, 1]])                                      # :48-52
    return state, F @ covariance @ F.T + MOTION_NOISE_3            # :55


def update3(state, covariance, observation, landmark_pos):
    dx = landmark_pos[0] - state[0]
    dy = landmark_pos[1] - state[1]
    q = dx ** 2 + dy ** 2
    zhat = np.array([np.sqrt(q), np.arctan2(dy, dx) - state[2]])
    H = np.array([[-dx / np.sqrt(q), -dy / np.sqrt(q), 0],
                  [dy / q, -dx / q, -1]])                          # :67-70
    S = H @ covariance @ H.T + OBSERVATION_NOISE_3                 # :73
    K = covariance @ H.T @ np.linalg.inv(S)                        # :74
    y = np.array(observation) - zhat
    y[1] = np.arctan2(np.sin(y[1]), np.cos(y[1]))                  # :79
    state = state + K @ y
    covariance = (np.eye(len(covariance)) - K @ H) @ covariance    # :83
    return state, covariance


# --------------------------------------------------------------------------
# STRUCTURED one-pass formulation = the specification of the HIP kernels
# --------------------------------------------------------------------------
def solve_compressed(mu_c: np.ndarray, p_cc: np.ndarray, lin, ang, ranges, bearings, cfg: EkfConfig):
    """The sequential part of one step on the compressed sub-state.

    C = [0,1,2, t_0,t_0+1, ...] (c = 3+2m entries).  Inputs are mu[C] and P[C,C]
    BEFORE prediction.  At iteration j the reference's loop (:436-480) needs from
    the current P only the rows sel_j (for H P) and the columns sel_j (for P H^T),
    sel_j = {0,1,2,t_j,t_j+1} a subset of C, so the recurrences close on the c x c
    system.  Rows and columns are kept distinct (K = P H^T S^-1 uses COLUMNS, :473;
    (I-KH)P uses ROWS, :480): replacing columns by rows through symmetry makes the
    rounding-level antisymmetric part of P grow exponentially.  Returns

      mu_c   (c,)      updated mean entries
      g      (2,)      G[0,2], G[1,2] of the motion Jacobian
      T      (2m, c)   V = T @ P'[C,:]   stacked rows H_j P_j           (2m x n)
      U      (c, 2m)   Kst = P'[:,C] @ U stacked gains K_j              (n x 2m)
      ys     (2m,)     stacked innovations y_j   (mean update is Kst @ ys)
    """
    m = len(ranges)
    c = 3 + 2 * m
    mu_c = np.array(mu_c, dtype=float)
    pose, G = motion_model(mu_c[0:3], lin, ang, cfg)
    if not cfg.disable_motion_model:
        mu_c[0:3] = pose
    g = np.array([G[0, 2], G[1, 2]])
    Gc = np.eye(c)
    Gc[0:3, 0:3] = G
    Pc = Gc @ p_cc @ Gc.T
    rd = cfg.motion_noise_diag()
    for i in range(3):
        Pc[i, i] += rd[i]
    A = np.eye(c)                     # current P[C,:] = A @ P'[C,:]
    B = np.eye(c)                     # current P[:,C] = P'[:,C] @ B
    T = np.zeros((2 * m, c))
    U = np.zeros((c, 2 * m))
    ys = np.zeros(2 * m)
    Qd = cfg.meas_noise_diag()
    if not cfg.enable_measurement_model:
        return mu_c, g, T, U, ys
    for j in range(m):
        a = 3 + 2 * j
        sel = [0, 1, 2, a, a + 1]
        y, h5 = innovation_and_h5(mu_c[0:3], mu_c[a:a + 2], ranges[j], bearings[j])
        hp = h5 @ Pc[sel, :]                       # (2,c)  H_j P_j at columns C
        ph = Pc[:, sel] @ h5.T                     # (c,2)  P_j H_j^T at rows C
        S = hp[:, sel] @ h5.T
        S[0, 0] += Qd[0]
        S[1, 1] += Qd[1]
        det = S[0, 0] * S[1, 1] - S[0, 1] * S[1, 0]
        Si = np.array([[S[1, 1], -S[0, 1]], [-S[1, 0], S[0, 0]]]) / det
        Kc = ph @ Si                               # (c,2)  K_j at rows C
        Tj = h5 @ A[sel, :]                        # (2,c)
        Uj = B[:, sel] @ h5.T @ Si                 # (c,2)
        mu_c = mu_c + Kc @ y
        Pc = Pc - Kc @ hp
        A = A - Kc @ Tj
        B = B - Uj @ hp
        T[2 * j:2 * j + 2] = Tj
        U[:, 2 * j:2 * j + 2] = Uj
        ys[2 * j:2 * j + 2] = y
    return mu_c, g, T, U, ys


def predicted_panels(cov: np.ndarray, C: np.ndarray, g: np.ndarray, rd: np.ndarray):
    """Row panel P'[C,:] (c x n) and column panel P'[:,C] (n x c) of P' = G_F P G_F^T + F^T R F (:430)."""
    R = cov[C, :].copy()
    R[0] += g[0] * cov[2]
    R[1] += g[1] * cov[2]
    r2 = R[:, 2].copy()
    R[:, 0] += g[0] * r2
    R[:, 1] += g[1] * r2
    L = cov[:, C].copy()
    L[0] += g[0] * cov[2, C]
    L[1] += g[1] * cov[2, C]
    l2 = L[:, 2].copy()
    L[:, 0] += g[0] * l2
    L[:, 1] += g[1] * l2
    for i in range(3):
        R[i, i] += rd[i]
        L[i, i] += rd[i]
    return R, L


def ekf_step_structured(mean, cov, lin, ang, idx, ranges, bearings, cfg: EkfConfig):
    """One predict+update step in O(n^2): compressed solve -> V,W panels -> one rank-K pass.

    P_new = P + Rt + sum_k W[:,k] V[k,:]  with K = 2m+2:
      k < 2m   : W = -P'[:,C] U,  V = T P'[C,:]   (the sequential updates, :480 unrolled)
      k = 2m   : W = gt,                V = P[2,:] + (P22/2) gt    (G_F P G_F^T, :430)
      k = 2m+1 : W = P[:,2]+(P22/2) gt, V = gt
    where gt = [g0, g1, 0, 0, ...] and Rt = diag(R) on the pose block.
    """
    mean = np.array(mean, dtype=float)
    cov = np.asarray(cov, dtype=float)
    n = len(mean)
    idx = [int(i) for i in idx]
    if not cfg.enable_measurement_model:
        idx, ranges, bearings = [], [], []
    m = len(idx)
    C = [0, 1, 2]
    for j in idx:
        C += [3 + 2 * j, 4 + 2 * j]
    C = np.array(C)
    mu_c, g, T, U, ys = solve_compressed(mean[C], cov[np.ix_(C, C)], lin, ang, ranges, bearings, cfg)
    gt = np.zeros(n)
    gt[0:2] = g
    rd = cfg.motion_noise_diag()
    R, L = predicted_panels(cov, C, g, rd)
    V = np.zeros((2 * m + 2, n))
    W = np.zeros((n, 2 * m + 2))
    V[:2 * m] = T @ R
    Kst = L @ U
    W[:, :2 * m] = -Kst
    p22 = cov[2, 2]
    V[2 * m] = cov[2, :] + 0.5 * p22 * gt
    W[:, 2 * m] = gt
    V[2 * m + 1] = gt
    W[:, 2 * m + 1] = cov[:, 2] + 0.5 * p22 * gt
    new_mean = mean + Kst @ ys
    new_mean[C] = mu_c
    new_cov = cov + W @ V
    for i in range(3):
        new_cov[i, i] += rd[i]
    return new_mean, new_cov


class DeferredSymmetricFilter:
    """NumPy restatement of the algebra the HIP kernels run (DESIGN.md "Formulation"), for the CPU tests.

    The covariance is symmetric and only its upper triangle is stored; pending low-rank terms are folded
    into the stored matrix once every few steps:

        P(a, b) = B[a, b] + sum_k W[a, k] V[k, b] + [a == b < 3] dacc[a]      for a <= b,   P(b, a) := P(a, b)

    One step (reference lines in src/replay_no_ros.py):
      * gather x[a, i] = P(C[a], i) for the c = 3+2m gathered rows C = [0,1,2, t_0,t_0+1, ...]  (k_solve / k_panels)
      * prediction P' = G_F P G_F^T + F^T R F (:430) changes rows 0,1 of the stored triangle only; those
        entries are added to B in place, the pose noise joins dacc
      * per observed landmark j (sequential, :436-480):  u = H_j P_j (rows from the panel),
        S = u[:, sel] h5^T + Q,  K_j = u^T S^-1  (P symmetric, so P H^T is the transpose of H P),
        mean += K_j y_j (:476),  panel -= K_j[C] u (:480),  appended ranks V = u, W = -K_j
      * flush: B[a, b] += W[a, :] V[:, b] on the upper triangle, B[a, a] += dacc[a]  (k_flush)
    The lower triangle of B is kept at NaN so that any read below the diagonal poisons the result.
    """

    def __init__(self, mean, cov_diag, cfg: EkfConfig, rank_limit: int = 80):
        n = len(mean)
        self.n, self.cfg, self.rank_limit = n, cfg, rank_limit
        self.mean = np.array(mean, dtype=float)
        self.B = np.full((n, n), np.nan)
        self.B[np.triu_indices(n)] = 0.0
        self.B[np.arange(n), np.arange(n)] = cov_diag
        self.W = np.zeros((n, 0))
        self.V = np.zeros((0, n))
        self.dacc = np.zeros(3)

    def _gather(self, C):
        ar = np.arange(self.n)
        X = np.empty((len(C), self.n))
        for a, ca in enumerate(C):
            lo, hi = np.minimum(ca, ar), np.maximum(ca, ar)
            X[a] = self.B[lo, hi]
            if self.W.shape[1]:
                X[a] += np.einsum("ik,ki->i", self.W[lo, :], self.V[:, hi])
            if ca < 3:
                X[a, ca] += self.dacc[ca]
        return X

    def flush(self):
        if self.W.shape[1]:
            iu = np.triu_indices(self.n)
            self.B[iu] += (self.W @ self.V)[iu]
        for a in range(3):
            self.B[a, a] += self.dacc[a]
        self.W, self.V, self.dacc = np.zeros((self.n, 0)), np.zeros((0, self.n)), np.zeros(3)

    def covariance(self):
        self.flush()
        up = np.triu(self.B)
        return up + np.triu(self.B, 1).T

    def step(self, lin, ang, idx, ranges, bearings):
        cfg, n = self.cfg, self.n
        idx = [int(j) for j in idx] if cfg.enable_measurement_model else []
        m = len(idx)
        if self.W.shape[1] + 2 * m > self.rank_limit:
            self.flush()
        C = [0, 1, 2]
        for j in idx:
            C += [3 + 2 * j, 4 + 2 * j]
        C = np.array(C)
        X = self._gather(C)
        pose, G = motion_model(self.mean[0:3], lin, ang, cfg)
        mu = self.mean.copy()
        if not cfg.disable_motion_model:
            mu[0:3] = pose
        g0, g1 = G[0, 2], G[1, 2]
        # prediction on the panel: row ops on rows 0,1, then the column ops of columns 0,1
        d0, d1 = g0 * X[2], g1 * X[2]
        col2 = X[:, 2].copy()
        col2[0] += g0 * X[2, 2]
        col2[1] += g1 * X[2, 2]
        d0[0] += g0 * col2[0]
        d0[1] += g1 * col2[0]
        d1[0] += g0 * col2[1]
        d1[1] += g1 * col2[1]
        X[2:, 0] += g0 * col2[2:]
        X[2:, 1] += g1 * col2[2:]
        X[0] += d0
        X[1] += d1
        self.B[0, :] += d0                                   # entries (0, i)
        self.B[1, 1:] += d1[1:]                              # entries (1, i), i >= 1
        rd = cfg.motion_noise_diag()
        for a in range(3):
            X[a, a] += rd[a]
        self.dacc += rd
        Qd = cfg.meas_noise_diag()
        Wn, Vn = np.zeros((n, 2 * m)), np.zeros((2 * m, n))
        for j in range(m):
            a = 3 + 2 * j
            sel = [0, 1, 2, a, a + 1]
            y, h5 = innovation_and_h5(mu[0:3], mu[C[a]:C[a] + 2], ranges[j], bearings[j])
            u = h5 @ X[sel, :]
            S = u[:, C[sel]] @ h5.T
            S[0, 0] += Qd[0]
            S[1, 1] += Qd[1]
            det = S[0, 0] * S[1, 1] - S[0, 1] * S[1, 0]
            Si = np.array([[S[1, 1], -S[0, 1]], [-S[1, 0], S[0, 0]]]) / det
            K = u.T @ Si
            mu = mu + K @ y
            X = X - K[C] @ u
            Vn[2 * j:2 * j + 2] = u
            Wn[:, 2 * j:2 * j + 2] = -K
        self.W = np.concatenate([self.W, Wn], axis=1)
        self.V = np.concatenate([self.V, Vn], axis=0)
        self.mean = mu


# --------------------------------------------------------------------------
# Synthetic stream of SURVEY.md section 8(d) / BASELINE.md section 3
# --------------------------------------------------------------------------
def synthetic_world(n_landmarks: int, trajectory_id: int = 0):
    """Landmarks uniform in a disc r=1.2 about (0,0.2); initial filter state."""
    rng = np.random.default_rng(1234 + trajectory_id)
    u = rng.random(n_landmarks)
    phi = rng.random(n_landmarks) * TWO_PI
    r = 1.2 * np.sqrt(u)
    lm = np.stack([r * np.cos(phi), 0.2 + r * np.sin(phi)], axis=1)
    mean0 = np.zeros(3 + 2 * n_landmarks)
    mean0[3:] = (lm + rng.normal(0.0, 0.05, lm.shape)).ravel()
    diag0 = np.full(3 + 2 * n_landmarks, 10000.0)
    diag0[0:3] = 0.1
    return rng, lm, mean0, diag0


def synthetic_stream(n_landmarks: int, steps: int, m: int = 8, trajectory_id: int = 0):
    """Returns (mean0, diag(P0), lin[steps], ang[steps], idx[steps,m], range[steps,m], bearing[steps,m]).

    Robot truth follows the reference motion model itself (motion_model above):
    lin = 0.004, ang = 0.02, every 10th step ang = 0.005 (straight branch, :376).
    """
    cfg = EkfConfig()
    rng, lm, mean0, diag0 = synthetic_world(n_landmarks, trajectory_id)
    lin = np.full(steps, 0.004)
    ang = np.full(steps, 0.02)
    ang[9::10] = 0.005
    idx = np.zeros((steps, m), dtype=np.int32)
    zr = np.zeros((steps, m))
    zb = np.zeros((steps, m))
    pose = np.zeros(3)
    for k in range(steps):
        pose, _ = motion_model(pose, lin[k], ang[k], cfg)
        vis = (m * k + np.arange(m)) % n_landmarks
        d = lm[vis] - pose[0:2]
        cth, sth = np.cos(pose[2]), np.sin(pose[2])
        xr = cth * d[:, 0] + sth * d[:, 1] + rng.normal(0.0, 0.01, m)
        yr = -sth * d[:, 0] + cth * d[:, 1] + rng.normal(0.0, 0.01, m)
        idx[k] = vis
        zr[k] = np.sqrt(xr ** 2 + yr ** 2)
        zb[k] = np.arctan2(yr, xr)
    return mean0, diag0, lin, ang, idx, zr, zb


def rel_fro(a: np.ndarray, b: np.ndarray) -> float:
    """Relative Frobenius distance |a-b|_F / |b|_F  (the parity metric, north_star: <= 1e-6)."""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / den) if den > 0 else float(np.linalg.norm(a - b))
