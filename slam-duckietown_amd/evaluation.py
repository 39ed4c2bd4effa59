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


def load_vicon_csv(path: str):
    """Rows of a Vicon export the way the reference reads them (scripts/decode_bag_file.py:139-146, :187-195):
    five header lines skipped, the rest split on commas.  Returns (fields, lengths): `fields` is (rows, width)
    float64 with NaN where a field is empty or absent, `lengths` the number of fields of every row."""
    with open(path, "r") as fh:
        for _ in range(5):
            fh.readline()
        rows = [ln.split(",") for ln in fh.read().strip().splitlines()]
    width = max(len(r) for r in rows)
    out = np.full((len(rows), width), np.nan)
    for i, r in enumerate(rows):
        for j, f in enumerate(r):
            if f != "":
                out[i, j] = float(f)
    return out, np.array([len(r) for r in rows])


def vicon_ground_truth(robot, markers, start_capture_time: float, end_capture_time: float, first_timestamp: float,
                       delay: float = 0.0, robot_len=None, marker_len=None) -> dict:
    """Ground truth of a Vicon recording in the frame the filter starts in: what the reference's
    ``get_ground_truth`` (scripts/decode_bag_file.py:111-253) turns into ``ground_truth`` / ``landmarks`` events.

    robot:   (rows, 8)  Frame, Sub Frame, RX, RY, RZ [rad], TX, TY, TZ [mm]            (:139-151)
    markers: (rows, 2 + 3 k)  Frame, Sub Frame, then X, Y, Z [mm] per marker, NaN = not seen   (:187-195)
    Times: frame / framerate + start_capture_time + delay with framerate = round(rows / capture length) (:149, :154).
    The Vicon frame closest to `first_timestamp` (the first event of the bag) defines the origin: every point is
    rotated about that position by that heading, the position is subtracted, mm -> m (:167-181, :241-246).  A marker's
    position is the middle of its x and y ranges over the recording (:201-222).  Kept from the reference so that its
    outputs are reproduced exactly: a coordinate equal to 0.0 counts as missing (`!= False`, :206, :214); a marker
    is dropped as a duplicate when its RAW position (mm) lies within 10 of an already accepted NORMALISED landmark
    (m), i.e. practically never (:231-239); a marker never seen sits at raw (0, 0); the "moved too much" test compares
    min - max (:225-228, the names are swapped at the unpacking) and so never fires.
    Raises ValueError where the reference prints and exits (:157-164).
    Returns dict(times, xy (rows, 2) [m], landmarks (k', 2) [m], landmarks_time, origin=(x, y, theta), framerate)."""
    robot = np.asarray(robot, dtype=float)
    markers = np.asarray(markers, dtype=float)
    n_rows = robot.shape[0]
    framerate = round(n_rows / (end_capture_time - start_capture_time))                      # :149
    ok = np.ones(n_rows, dtype=bool) if robot_len is None else np.asarray(robot_len) == 8    # :148
    rob = robot[ok]
    times = rob[:, 0].astype(np.int64) / framerate + start_capture_time + delay             # :154
    k = int(np.argmin(np.abs(times - first_timestamp)))                                      # :157 (stable: first minimum)
    if abs(times[k] - first_timestamp) > 2:                                                  # :158
        raise ValueError("cannot find a matching time between the ground truth and the bag data "
                         f"(closest differs by {times[k] - first_timestamp:.3f} s); use `delay`")
    init_theta, init_x, init_y = float(rob[k, 4]), float(rob[k, 5]), float(rob[k, 6])      # :168
    c, s = math.cos(init_theta), math.sin(init_theta)

    def normalise(x, y):                                                                     # rotate_around + :177-178
        rx = init_x + (x - init_x) * c - (y - init_y) * s
        ry = init_y + (x - init_x) * s + (y - init_y) * c
        return (rx - init_x) / 1000, (ry - init_y) / 1000

    gx, gy = normalise(rob[:, 5], rob[:, 6])
    mok = np.ones(markers.shape[0], dtype=bool) if marker_len is None else np.asarray(marker_len) > 4    # :192
    mk = markers[mok]
    count = mk.shape[1] // 3                                                                 # :196
    landmarks = []
    for q in range(count):
        xs, ys = mk[:, 3 * q + 2], mk[:, 3 * q + 3]
        xs, ys = xs[~np.isnan(xs) & (xs != 0.0)], ys[~np.isnan(ys) & (ys != 0.0)]          # `!= False`
        hi_x, lo_x = (xs.max(), xs.min()) if xs.size else (0.0, 0.0)
        hi_y, lo_y = (ys.max(), ys.min()) if ys.size else (0.0, 0.0)
        if lo_x - hi_x > 20 or lo_y - hi_y > 20:                                             # :225-228 as written
            raise ValueError("landmarks have moved too much")
        pos_x, pos_y = (hi_x + lo_x) / 2, (hi_y + lo_y) / 2                                  # :230
        if any((pos_x - lx) ** 2 + (pos_y - ly) ** 2 < 10 ** 2 for lx, ly in landmarks):    # :233-239
            continue
        nx, ny = normalise(pos_x, pos_y)
        landmarks.append((float(nx), float(ny)))
    return dict(times=times, xy=np.stack([gx, gy], axis=1), landmarks=np.array(landmarks, dtype=float).reshape(-1, 2),
                landmarks_time=first_timestamp, origin=(init_x, init_y, init_theta), framerate=framerate)


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
