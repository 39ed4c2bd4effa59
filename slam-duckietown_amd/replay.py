"""Offline replay driver for the reference's ``events.csv`` wire format.

Writer of the format: ``scripts/decode_bag_file.py:62-105, 337-347`` (one ``timestamp,event,data`` line per
event, sorted by timestamp).  Reader this module mirrors: ``src/replay_no_ros.py:66-248``:

* ``left_wheel`` / ``right_wheel``  encoder ticks are latched (:103-118);
* ``image``                         AprilTags of the frame join the current window (:120-131);
* ``ground_truth``                  ``x,y`` of the Vicon track (:99-101);
* ``landmarks``                     literal list of true tag positions; with ``god_key`` the filter is
                                    pre-sized with them like ``ENABLE_GOD_EKF`` (:133-157);
* ``camera_intrinsis``              literal ``[D, K, R, P]``; ``K`` gives ``[fx, fy, cx, cy]`` (:160-182);
* whenever an event arrives more than ``delta_time`` after the window start, ONE EKF step runs on the
  odometry of the latched ticks and the window's detections, and the window start advances by exactly
  ``delta_time`` (:194-238).

One extension: a ``detections`` event carrying ``[(tag_id, [tx, ty, tz], pose_err), ...]`` lets a log be
replayed without ``dt_apriltags`` / ``cv2`` (neither is needed by this package).  ``image`` events need a
``detector(path, camera_params) -> [tag, ...]`` callable.

The EKF itself runs on the GPU (``GpuBackend`` -> ``EkfSlam``); the driver only parses, windows and
integrates ticks.  Unknown events raise ``ValueError`` (the reference prints and exits, :185-188).
"""
from __future__ import annotations

import ast
import dataclasses
import os
from types import SimpleNamespace
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from .frontend import associate, delta_phi, displacement


@dataclasses.dataclass
class ReplayResult:
    poses: List[np.ndarray]                 # pose after every window (x, y, theta)
    path: List[tuple]                       # (x, y) trail starting at (0, 0), like acc_pos (:83, :245)
    mean: np.ndarray                        # final state
    covariance: np.ndarray
    tag_index: Dict[int, int]
    measured_tags: List[dict]               # tags_positions of every window
    ground_truth: List[tuple]
    landmarks: Optional[list]
    camera_params: list
    windows: int


class GpuBackend:
    """Runs association on the host and augmentation + predict + update on the MI355X."""

    def __init__(self, config=None, capacity: int = 79, device: int = 0, device_association: bool = False):
        """capacity: n_max of the first handle (79 = 38 landmarks: the small-state path, one launch per window -- the
        reference's map has 12, src/replay_no_ros.py:26); the handle is re-created with twice the room when the map outgrows it.
        device_association=True: association / gate / averaging / augmentation run on the GPU too
        (`EkfSlam.step_detections`).  The device front end has limits the reference does not (16 distinct tags
        and 64 detections per window, tag ids below 1024, a fixed capacity): every window is checked against them
        on the host first; the state is regrown when the map would not fit, and a window beyond the per-window
        limits takes the host association (which has none) for that window."""
        from .ekf_bindings import EkfConfig, EkfSlam
        self.config = config or EkfConfig()
        self.device_association = device_association
        self._make = lambda cap: EkfSlam(cap | 1, 1, device, self.config)
        self.filt = self._make(capacity)
        self._cache = None            # (mean, covariance) of the device state when the last step brought them back with it
        self._dev_index = None        # device_association: the device's tag table as last downloaded (one download per window)
        if device_association:
            self.filt.set_association(self.config.gate_range, self.config.ignore_tags)

    def set_state(self, mean, cov):
        self._cache = None
        self._dev_index = None
        n = len(mean)
        if n > self.filt.n_max:
            self.filt.close()
            self.filt = self._make(max(n, 2 * self.filt.n_max - 3))
            if self.device_association:
                self.filt.set_association(self.config.gate_range, self.config.ignore_tags)
        self.filt.set_state(mean, cov)

    def _grow(self, n_needed):
        self._cache = None
        self._dev_index = None
        mean, cov = self.filt.state()
        index = self.filt.tag_index() if self.device_association else None
        self.filt.close()
        self.filt = self._make(max(n_needed, 2 * self.filt.n_max - 3))
        self.filt.set_state(mean, cov)
        if self.device_association:
            self.filt.set_association(self.config.gate_range, self.config.ignore_tags)
            if index:
                self.filt.set_tag_index(index)

    def pose(self) -> np.ndarray:
        if self._cache is not None:
            return self._cache[0][:3].copy()
        return self.filt.mean()[:3]

    def state(self):
        if self._cache is not None:
            return self._cache[0].copy(), self._cache[1].copy()
        return self.filt.state()

    def _window_tags(self, detections):
        """Tag ids of the window that pass IGNORE_TAGS and the gate (:286-289), in order of first appearance."""
        ids, count = [], 0
        gate2 = self.config.gate_range ** 2
        for _stamp, tags in detections:
            for tag in tags:
                count += 1
                t = np.asarray(tag.pose_t, dtype=np.float64).ravel()
                if tag.tag_id in self.config.ignore_tags or t[2] ** 2 + t[0] ** 2 > gate2:
                    continue
                if tag.tag_id not in ids:
                    ids.append(tag.tag_id)
        return ids, count

    def step(self, ang, lin, detections, tag_index) -> dict:
        from .ekf_bindings import EKF_AMAX, EKF_DMAX, EKF_TAGMAX
        cached, self._cache = self._cache, None
        if self.device_association:
            dev_index = dict(self._dev_index) if self._dev_index is not None else self.filt.tag_index()
            if tag_index and not dev_index:                       # a pre-filled TAG_INDEX (god mode) goes to the device
                self.filt.set_tag_index(tag_index)
                dev_index = dict(tag_index)
            ids, count = self._window_tags(detections)
            n_need = 3 + 2 * (len(dev_index) + sum(1 for t in ids if t not in dev_index))
            if n_need > self.filt.n_max:
                self._grow(n_need)
            if len(ids) <= EKF_AMAX and count <= EKF_DMAX and all(0 <= t < EKF_TAGMAX for t in ids):
                self.filt.step_detections(lin, ang, detections)
                tags = self.filt.tags_positions()                 # raises if the device still had to drop something
                self._dev_index = self.filt.tag_index()
                tag_index.clear()
                tag_index.update(self._dev_index)
                return tags
            tag_index.clear()                                     # this window: host association, no limits
            tag_index.update(dev_index)
        pose = cached[0][:3] if cached is not None else self.pose()   # world guesses use the pre-step pose (:331-332)
        tags = associate(detections, tag_index, pose, self.config.gate_range, self.config.ignore_tags)
        n_old = self.filt.size()
        n_new = max(n_old, 3 + 2 * len(tag_index))
        if n_new > self.filt.n_max:
            self._grow(n_new)
        new_xy = [(tags[(i - 3) // 2][0], tags[(i - 3) // 2][1]) for i in range(n_old, n_new, 2)]   # KeyError like :359
        if new_xy:
            self.filt.add_landmarks(np.array(new_xy))
        idx = list(tags.keys())
        if self.filt.n_max <= 131:
            # a small state comes back with the step (ekf_step_fetch: on the small-state path ONE launch, whose result the
            # host polls for): the loop reads the pose after every window (:241) and the next window's association needs it
            self._cache = self.filt.step_state(lin, ang, idx, [tags[k][4] for k in idx], [tags[k][5] for k in idx])
        else:
            self.filt.step(lin, ang, idx, [tags[k][4] for k in idx], [tags[k][5] for k in idx])
        self._dev_index = None
        if self.device_association and all(0 <= t < EKF_TAGMAX for t in tag_index):
            self.filt.set_tag_index(tag_index)                    # the device table follows the host's for the next window
            self._dev_index = dict(tag_index)
        return tags

    def close(self):
        self.filt.close()


def _tags_from_literal(items):
    tags = []
    for tag_id, t, err in items:
        tags.append(SimpleNamespace(tag_id=int(tag_id), pose_R=np.eye(3),
                                    pose_t=np.asarray(t, dtype=float).reshape(3, 1), pose_err=float(err)))
    return tags


def _payload(line: str) -> str:
    first = line.find(",")
    second = line.find(",", first + 1)
    return line[second + 1:]


def replay(source, backend=None, delta_time: float = 0.7, detector: Optional[Callable] = None,
           god_key: Optional[Sequence[int]] = None, wheel_radius: float = 0.0318, baseline: float = 0.1,
           resolution: int = 135, motion_sigma: float = 0.1, on_window: Optional[Callable] = None,
           fast_mode: bool = False) -> ReplayResult:
    """Replay an ``events.csv`` (directory, file path, or an iterable of lines) through the EKF.

    ``backend`` defaults to ``GpuBackend()``; it needs ``set_state / pose / state / step`` (tests inject a
    CPU double).  ``on_window(result_so_far_dict)`` is called after every EKF step (the reference plots there).
    ``fast_mode`` is the reference's ``ENABLE_FAST_MODE`` (:32): frames are only queued when they arrive
    (:122-123) and, at every window boundary, the detector runs on the LAST FIVE frames queued so far -- the
    queue is never emptied, so a window with fewer than five frames reaches back into earlier ones (:216-227).
    """
    base_dir = None
    if isinstance(source, (str, os.PathLike)):
        path = os.fspath(source)
        if os.path.isdir(path):
            base_dir, path = path, os.path.join(path, "events.csv")
        else:
            base_dir = os.path.dirname(path)
        with open(path, "r") as fh:
            lines = fh.readlines()
    else:
        lines = list(source)
    own_backend = backend is None
    if backend is None:
        backend = GpuBackend()
    backend.set_state(np.array([0.0, 0.0, 0.0]), np.eye(3) * motion_sigma)        # :69-70

    prev_ltick = prev_rtick = curr_ltick = curr_rtick = False                    # :74-78
    prev_stamp = False                                                           # :81
    tag_index: Dict[int, int] = {}
    window: list = []
    image_list: list = []                                                        # fast mode: every frame so far (:123)
    poses, path_xy, measured = [], [(0, 0)], []
    ground_truth, landmarks = [], None
    camera_params = [340, 336, 328, 257]                                         # :93
    windows = 0
    try:
        for line in lines:
            if not line.strip():
                continue
            stamp_s, event, data, *rest = line.strip().split(",")
            if event == "ground_truth":
                ground_truth.append((float(data), float(rest[0])))
            elif event == "left_wheel":
                ticks = int(data)
                if prev_ltick == False:                                          # noqa: E712  (0 re-latches, like :105)
                    prev_ltick = ticks
                curr_ltick = ticks
            elif event == "right_wheel":
                ticks = int(data)
                if prev_rtick == False:                                          # noqa: E712
                    prev_rtick = ticks
                curr_rtick = ticks
            elif event == "image":
                if detector is None:
                    raise ValueError("'image' event needs a detector(path, camera_params) callable")
                img_path = os.path.join(base_dir, data) if base_dir else data
                if fast_mode:
                    image_list.append((stamp_s, lambda cp, p=img_path: detector(p, cp)))
                else:
                    window.append((stamp_s, detector(img_path, camera_params)))
            elif event == "detections":
                tags_now = _tags_from_literal(ast.literal_eval(_payload(line.strip())))
                if fast_mode:
                    image_list.append((stamp_s, lambda cp, t=tags_now: t))
                else:
                    window.append((stamp_s, tags_now))
            elif event == "landmarks":
                landmarks = ast.literal_eval(_payload(line.strip()))
                if god_key is not None:                                          # ENABLE_GOD_EKF, :140-157
                    mean, cov = backend.state()
                    size = len(landmarks) * 2 + 3
                    new_mean = np.zeros(size)
                    new_mean[0:2] = mean[0:2]
                    for i, lm in enumerate(landmarks):
                        new_mean[3 + 2 * i], new_mean[4 + 2 * i] = lm[0], lm[1]
                    new_cov = np.zeros((size, size))
                    new_cov[0:3, 0:3] = cov[0:3, 0:3]
                    backend.set_state(new_mean, new_cov)
                    tag_index = {god_key[i]: i for i in range((size - 3) // 2)}
            elif event == "camera_intrinsis":
                K = ast.literal_eval(_payload(line.strip()))[1]
                camera_params = [K[0], K[4], K[2], K[5]]
            else:
                raise ValueError(f"unknown event {event!r}")

            stamp = float(stamp_s)
            if prev_stamp == False:                                              # noqa: E712  (:191)
                prev_stamp = stamp
            if stamp - prev_stamp > delta_time:                                  # :196
                windows += 1
                prev_stamp = prev_stamp + delta_time                             # :199
                d_l = delta_phi(curr_ltick, prev_ltick, resolution)
                d_r = delta_phi(curr_rtick, prev_rtick, resolution)
                prev_ltick, prev_rtick = curr_ltick, curr_rtick
                ang, lin = displacement(wheel_radius, baseline, d_l, d_r)
                if fast_mode:                                                    # :216-227
                    window = [(stamp, detect(camera_params)) for _stamp, detect in image_list[-5:]]
                tags = backend.step(ang, lin, window, tag_index)
                window = []                                                      # :238
                pose = np.array(backend.pose(), dtype=float)
                poses.append(pose)
                path_xy.append((pose[0], pose[1]))
                measured.append(tags)
                if on_window is not None:
                    on_window(dict(window=windows, pose=pose, tags=tags, tag_index=tag_index))
        mean, cov = backend.state()
        return ReplayResult(poses, path_xy, mean, cov, tag_index, measured, ground_truth, landmarks,
                            camera_params, windows)
    finally:
        if own_backend:
            backend.close()
