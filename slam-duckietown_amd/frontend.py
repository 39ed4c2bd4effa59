"""Host-side front end of one EKF step: what stays on the CPU.

Association / range gate / averaging of the AprilTag detections of one window, and the wheel
odometry scalars.  O(#detections) Python, mirrors the reference's semantics exactly:

  associate      src/replay_no_ros.py:280-337
  delta_phi      src/replay_no_ros.py:250-266
  displacement   src/replay_no_ros.py:484-497
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np


def delta_phi(ticks: int, prev_ticks: int, resolution: int) -> float:
    """Wheel rotation in radians for an encoder tick difference (replay_no_ros.py:262-264)."""
    return (ticks - prev_ticks) * (2 * np.pi / resolution)


def displacement(R: float, baseline: float, delta_phi_left: float, delta_phi_right: float) -> Tuple[float, float]:
    """(angular, linear) displacement of a differential drive (replay_no_ros.py:491-497)."""
    right = R * delta_phi_right
    left = R * delta_phi_left
    return (right - left) / baseline, (left + right) / 2


def associate(detections, tag_index: Dict[int, int], pose, gate_range: float = 1.5,
              ignore_tags: Sequence[int] = ()):
    """Tag id -> landmark index, 1.5 m gate, per-tag averaging over the window's frames.

    ``detections`` is the reference's ``[(timestamp, [tag, ...])]`` list; a tag needs ``tag_id``,
    ``pose_t`` (3,1) and ``pose_err``.  ``tag_index`` is mutated (new ids get the next index,
    :294-295).  Returns ``{landmark_index: [xw, yw, err, tag_id, range, bearing]}`` in order of
    first appearance -- which is the order the update processes them in (:436).
    The world-frame guess uses ``pose`` BEFORE the prediction (:331-332).
    """
    # Sums instead of lists of arrays: `np.mean(list_of_(3,1)_arrays, axis=0)` (:315) adds the frames in order and divides by
    # their number -- the same additions, in the same order, on Python floats (only pose_t[0] and pose_t[2] are ever used,
    # :321); the errors are averaged the same way up to 7 detections (NumPy sums short 1-D arrays sequentially; from 8 on it
    # sums pairwise, and np.mean itself is called to stay bit-identical).  What costs time here is Python, not arithmetic:
    # this function is a third of a drop-in call at the reference's map size.
    gate2 = gate_range ** 2
    seen: Dict[int, List] = {}
    for _stamp, tags in detections:
        for tag in tags:
            tid = tag.tag_id
            if tid in ignore_tags:
                continue
            t = tag.pose_t
            tx, tz = float(t[0][0]), float(t[2][0])
            if tz * tz + tx * tx > gate2:
                continue
            lm = tag_index.get(tid)
            if lm is None:
                lm = tag_index[tid] = len(tag_index)
            acc = seen.get(lm)
            if acc is None:
                seen[lm] = [tx, tz, [tag.pose_err], tid]
            else:
                acc[0] += tx
                acc[1] += tz
                acc[2].append(tag.pose_err)
    x0, y0, th = float(pose[0]), float(pose[1]), float(pose[2])
    result = {}
    for lm, (sx, sz, errs, tid) in seen.items():
        k = len(errs)
        if k < 8:
            e = errs[0]
            for v in errs[1:]:
                e = e + v
            err = np.float64(e) / k
        else:
            err = np.mean(errs, axis=0)
        x_r, y_r = np.float64(sz / k), np.float64(-(sx / k))
        rng = np.sqrt(x_r ** 2 + y_r ** 2)
        brg = np.arctan2(y_r, x_r)
        result[lm] = [x0 + rng * np.cos(brg + th), y0 + rng * np.sin(brg + th), err, tid, rng, brg]
    return result
