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
    seen: Dict[int, List] = {}
    for _stamp, tags in detections:
        for tag in tags:
            if tag.tag_id in ignore_tags:
                continue
            if tag.pose_t[2][0] ** 2 + tag.pose_t[0][0] ** 2 > gate_range ** 2:
                continue
            if tag.tag_id not in tag_index:
                tag_index[tag.tag_id] = len(tag_index)
            seen.setdefault(tag_index[tag.tag_id], []).append((tag.pose_t, tag.pose_err))
    id_of = {v: k for k, v in tag_index.items()}
    x0, y0, th = float(pose[0]), float(pose[1]), float(pose[2])
    result = {}
    for lm, obs in seen.items():
        t = np.mean([o[0] for o in obs], axis=0)
        err = np.mean([o[1] for o in obs], axis=0)
        x_r, y_r = t[2][0], -t[0][0]
        rng = np.sqrt(x_r ** 2 + y_r ** 2)
        brg = np.arctan2(y_r, x_r)
        result[lm] = [x0 + rng * np.cos(brg + th), y0 + rng * np.sin(brg + th), err, id_of[lm], rng, brg]
    return result
