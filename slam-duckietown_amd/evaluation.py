"""Accuracy and consistency evaluation.  Host-side NumPy; the filters run on the GPU.

* Vicon alignment of the reference's ground truth (scripts/decode_bag_file.py:107-109, :167-181, :241-246):
  rotate the Vicon points about the robot's first pose by its first heading, subtract that pose, mm -> m.
* ATE / RMSE of an estimated path against ground truth (what the reference only eyeballs in its plots,
  src/replay_no_ros.py:520-529).
* NEES of the pose over a batch of Monte-Carlo trajectories with chi-square consistency bounds: the
  batched filter bank (`EkfSlam(batch=B)`) is exactly the Monte-Carlo tool this needs.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import numpy as np


def rotate_around(a, b, x, y, theta):
    """Rotate (x, y) about (a, b) by theta (scripts/decode_bag_file.py:107-109)."""
    return (a + (x - a) * math.cos(theta) - (y - b) * math.sin(theta),
            b + (x - a) * math.sin(theta) + (y - b) * math.cos(theta))


def align_vicon(points_mm, init_x: float, init_y: float, init_theta: float) -> np.ndarray:
    """Vicon points (mm, Vicon frame) -> metres in the frame the reference plots them in (:176-178, :242-244)."""
    pts = np.asarray(points_mm, dtype=float).reshape(-1, 2)
    c, s = math.cos(init_theta), math.sin(init_theta)
    dx, dy = pts[:, 0] - init_x, pts[:, 1] - init_y
    x = init_x + dx * c - dy * s
    y = init_y + dx * s + dy * c
    return np.stack([(x - init_x) / 1000.0, (y - init_y) / 1000.0], axis=1)


def wrap_angle(a):
    return (np.asarray(a) + np.pi) % (2 * np.pi) - np.pi


def ate_rmse(estimate_xy, truth_xy, align: bool = False) -> float:
    """Absolute trajectory error (RMSE of the position difference).  align=True first removes the best
    rigid transform (rotation + translation, Kabsch) -- use it when the two frames are not registered."""
    e = np.asarray(estimate_xy, dtype=float).reshape(-1, 2)
    t = np.asarray(truth_xy, dtype=float).reshape(-1, 2)
    if e.shape != t.shape:
        raise ValueError("estimate and truth must have the same number of points")
    if align:
        ec, tc = e.mean(axis=0), t.mean(axis=0)
        H = (e - ec).T @ (t - tc)
        U, _, Vt = np.linalg.svd(H)
        d = np.sign(np.linalg.det(Vt.T @ U.T))
        R = Vt.T @ np.diag([1.0, d]) @ U.T
        e = (e - ec) @ R.T + tc
    return float(np.sqrt(np.mean(np.sum((e - t) ** 2, axis=1))))


def resample_truth(truth_t, truth_xy, query_t) -> np.ndarray:
    """Ground-truth positions linearly interpolated at the estimate's time stamps."""
    truth_t = np.asarray(truth_t, dtype=float)
    truth_xy = np.asarray(truth_xy, dtype=float).reshape(-1, 2)
    return np.stack([np.interp(query_t, truth_t, truth_xy[:, 0]), np.interp(query_t, truth_t, truth_xy[:, 1])], axis=1)


def nees(errors, covariances) -> np.ndarray:
    """e^T P^-1 e per sample; errors (B, d), covariances (B, d, d)."""
    e = np.asarray(errors, dtype=float)
    P = np.asarray(covariances, dtype=float)
    return np.einsum("bi,bi->b", e, np.linalg.solve(P, e[..., None])[..., 0])


def chi2_bounds(dof: int, runs: int, confidence: float = 0.95) -> Tuple[float, float]:
    """Two-sided bounds for the AVERAGE NEES of `runs` independent runs (Bar-Shalom's ANEES test)."""
    from scipy.stats import chi2
    a = (1.0 - confidence) / 2.0
    return chi2.ppf(a, dof * runs) / runs, chi2.ppf(1.0 - a, dof * runs) / runs


def pose_nees(filter_bank, true_poses: Sequence[Sequence[float]]):
    """NEES of [x, y, theta] for every trajectory of an ``EkfSlam`` bank; downloads only the 3x3 pose blocks.

    Returns (nees per trajectory (B,), average NEES, (lower, upper) 95 % bounds for a consistent filter).
    An average above the upper bound means the filter is over-confident, below the lower one conservative.
    """
    true_poses = np.asarray(true_poses, dtype=float).reshape(filter_bank.batch, 3)
    errs, covs = [], []
    for b in range(filter_bank.batch):
        mu = filter_bank.mean(b)[:3]
        e = mu - true_poses[b]
        e[2] = wrap_angle(e[2])
        errs.append(e)
        covs.append(filter_bank.covariance_block(0, 0, 3, 3, b))
    vals = nees(np.array(errs), np.array(covs))
    return vals, float(vals.mean()), chi2_bounds(3, filter_bank.batch)
