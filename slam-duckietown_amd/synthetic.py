"""Synthetic odometry + landmark streams of SURVEY.md 8(d) / BASELINE.md section 3 (benchmark inputs).

Seeded per trajectory (``np.random.default_rng(1234 + trajectory_id)``): N landmarks uniform in a disc
of radius 1.2 m about (0, 0.2); the robot follows the reference's own kinematics
(src/replay_no_ros.py:376-397) with lin = 0.004, ang = 0.02 and every 10th step ang = 0.005 (straight
branch); step k observes landmarks (m*k + i) mod N with N(0, 0.01^2) noise on the robot-frame position.
"""
from __future__ import annotations

import numpy as np


def _advance(pose, lin, ang):
    x, y, th = pose
    if abs(ang) <= 1e-2:
        return np.array([x + lin * np.cos(th), y + lin * np.sin(th), th])
    r = lin / ang
    th2 = (th + ang + np.pi) % (2 * np.pi) - np.pi
    return np.array([x + (-r * np.sin(th) + r * np.sin(th + ang)), y + (r * np.cos(th) - r * np.cos(th + ang)), th2])


def true_poses(steps: int) -> np.ndarray:
    """The robot's true pose after each step of synthetic_stream (identical for every trajectory id)."""
    lin = np.full(steps, 0.004)
    ang = np.full(steps, 0.02)
    ang[9::10] = 0.005
    pose = np.zeros(3)
    out = np.zeros((steps, 3))
    for k in range(steps):
        pose = _advance(pose, lin[k], ang[k])
        out[k] = pose
    return out


def _world(n_landmarks: int, trajectory_id: int):
    rng = np.random.default_rng(1234 + trajectory_id)
    u = rng.random(n_landmarks)
    phi = rng.random(n_landmarks) * 2 * np.pi
    r = 1.2 * np.sqrt(u)
    lm = np.stack([r * np.cos(phi), 0.2 + r * np.sin(phi)], axis=1)
    mean0 = np.zeros(3 + 2 * n_landmarks)
    mean0[3:] = (lm + rng.normal(0.0, 0.05, lm.shape)).ravel()
    diag0 = np.full(3 + 2 * n_landmarks, 10000.0)
    diag0[0:3] = 0.1
    return rng, lm, mean0, diag0


def variable_stream(n_landmarks: int, steps: int, m_lo: int = 0, m_hi: int = 8, trajectory_id: int = 0):
    """The shapes the reference's loop produces (src/replay_no_ros.py:280-301, :436: whatever tags the window saw): per
    step m ~ uniform{m_lo..m_hi} landmarks drawn WITHOUT order or locality (``rng.choice(N, m, replace=False)``).
    Same world, kinematics and noise as synthetic_stream.
    -> mean0, diagP0, lin[steps], ang[steps], idx[steps, m_hi] (entries beyond m[k] are 0), range, bearing, m[steps] int32."""
    rng, lm, mean0, diag0 = _world(n_landmarks, trajectory_id)
    lin = np.full(steps, 0.004)
    ang = np.full(steps, 0.02)
    ang[9::10] = 0.005
    stride = max(1, m_hi)
    idx = np.zeros((steps, stride), dtype=np.int32)
    zr = np.zeros((steps, stride))
    zb = np.zeros((steps, stride))
    mk = rng.integers(m_lo, m_hi + 1, size=steps).astype(np.int32)
    pose = np.zeros(3)
    for k in range(steps):
        pose = _advance(pose, lin[k], ang[k])
        m = int(mk[k])
        vis = rng.choice(n_landmarks, m, replace=False)
        d = lm[vis] - pose[0:2]
        c, s = np.cos(pose[2]), np.sin(pose[2])
        xr = c * d[:, 0] + s * d[:, 1] + rng.normal(0.0, 0.01, m)
        yr = -s * d[:, 0] + c * d[:, 1] + rng.normal(0.0, 0.01, m)
        idx[k, :m] = vis
        zr[k, :m] = np.sqrt(xr ** 2 + yr ** 2)
        zb[k, :m] = np.arctan2(yr, xr)
    return mean0, diag0, lin, ang, idx, zr, zb, mk


def synthetic_stream(n_landmarks: int, steps: int, m: int = 8, trajectory_id: int = 0):
    """-> mean0 (n,), diagP0 (n,), lin[steps], ang[steps], idx[steps,m] int32, range[steps,m], bearing[steps,m]."""
    rng = np.random.default_rng(1234 + trajectory_id)
    u = rng.random(n_landmarks)
    phi = rng.random(n_landmarks) * 2 * np.pi
    r = 1.2 * np.sqrt(u)
    lm = np.stack([r * np.cos(phi), 0.2 + r * np.sin(phi)], axis=1)
    mean0 = np.zeros(3 + 2 * n_landmarks)
    mean0[3:] = (lm + rng.normal(0.0, 0.05, lm.shape)).ravel()
    diag0 = np.full(3 + 2 * n_landmarks, 10000.0)
    diag0[0:3] = 0.1
    lin = np.full(steps, 0.004)
    ang = np.full(steps, 0.02)
    ang[9::10] = 0.005
    idx = np.zeros((steps, m), dtype=np.int32)
    zr = np.zeros((steps, m))
    zb = np.zeros((steps, m))
    pose = np.zeros(3)
    for k in range(steps):
        pose = _advance(pose, lin[k], ang[k])
        vis = (m * k + np.arange(m)) % n_landmarks
        d = lm[vis] - pose[0:2]
        c, s = np.cos(pose[2]), np.sin(pose[2])
        xr = c * d[:, 0] + s * d[:, 1] + rng.normal(0.0, 0.01, m)
        yr = -s * d[:, 0] + c * d[:, 1] + rng.normal(0.0, 0.01, m)
        idx[k] = vis
        zr[k] = np.sqrt(xr ** 2 + yr ** 2)
        zb[k] = np.arctan2(yr, xr)
    return mean0, diag0, lin, ang, idx, zr, zb
