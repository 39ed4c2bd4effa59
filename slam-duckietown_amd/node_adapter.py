"""ROS-free adapter for the online node.

``packages/histogram_lane_filter/src/histogram_lane_filter_node.py`` runs the EKF from a 30 Hz timer
(``cbPredict`` :164-217) fed by two encoder callbacks (:148-162) and the camera callback (``cbImage``
:219-254).  This class is that EKF part with the ROS types stripped off, so a node only forwards its
messages:

    self.ekf = EkfNodeAdapter()                                   # in __init__
    self.ekf.on_left_encoder(msg.data, msg.resolution)            # cbProcessLeftEncoder
    self.ekf.on_right_encoder(msg.data, msg.resolution)           # cbProcessRightEncoder
    self.ekf.on_image(rospy.Time.now().to_nsec(), detected_tags)  # cbImage, after detector.detect(...)
    pose = self.ekf.on_timer()                                    # cbPredict (None: encoders did not move)

Two deliberate differences from the reference node, both fixes of latent bugs it documents in SURVEY.md:
the tag-id -> landmark-index map persists between ticks (the node passes ``TAG_INDEX={}`` at :198, which
restarts indices every tick; ``persistent_tag_index=False`` restores that), and the detection list shared by
the camera and timer threads is guarded by a lock (the node appends at :242 and replaces at :205 unlocked).

Parity: the arithmetic per tick is ``delta_phi(delta_ticks, 0, resolution)`` (:183-184), ``displacement`` (:186)
and one ``EKF_pose_estimation`` call (:197-199), all pinned elsewhere; the callback sequencing itself cannot be
pinned against the reference (it needs rospy) and is covered by ``tests/test_node_adapter.py`` against a
restatement of :148-217.
"""
from __future__ import annotations

import threading
from typing import Optional

import numpy as np

from .frontend import delta_phi, displacement


class EkfNodeAdapter:
    def __init__(self, backend=None, wheel_radius: float = 0.0318, baseline: float = 0.1,
                 persistent_tag_index: bool = True, device_association: bool = False):
        if backend is None:
            from .replay import GpuBackend
            backend = GpuBackend(device_association=device_association)
        self.backend = backend
        self.backend.set_state(np.array([0.0, 0.0, 0.0]), np.eye(3) * 0.1)       # node :56-57
        self.wheel_radius, self.baseline = wheel_radius, baseline                # :180-181
        self.persistent_tag_index = persistent_tag_index
        self.tag_index = {}
        self.resolution = None                        # filter.encoder_resolution, set by the first encoder message
        self.left_ticks = self.right_ticks = 0        # :62-63
        self.left_delta = self.right_delta = 0        # :50-51
        self.detections = []
        self.tags = {}
        self.path = []                                # acc_pos
        self._lock = threading.Lock()

    def on_left_encoder(self, ticks: int, resolution: int):
        if self.resolution is None:
            self.resolution = resolution
        self.left_delta = ticks - self.left_ticks                                # :152

    def on_right_encoder(self, ticks: int, resolution: int):
        if self.resolution is None:
            self.resolution = resolution
        self.right_delta = ticks - self.right_ticks                              # :159-161

    def on_image(self, stamp, detected_tags):
        if detected_tags:                                                        # :240-242
            with self._lock:
                self.detections.append((stamp, list(detected_tags)))

    def on_timer(self) -> Optional[np.ndarray]:
        if self.right_delta == 0 and self.left_delta == 0:                       # :166-167
            return None
        self.left_ticks += self.left_delta                                       # :175-176
        self.right_ticks += self.right_delta
        d_l = delta_phi(self.left_delta, 0, self.resolution)                     # :183-184
        d_r = delta_phi(self.right_delta, 0, self.resolution)
        ang, lin = displacement(self.wheel_radius, self.baseline, d_l, d_r)      # :186
        self.left_delta = self.right_delta = 0                                   # :189-190
        with self._lock:
            window, self.detections = self.detections, []                        # :205
        tag_index = self.tag_index if self.persistent_tag_index else {}
        tags = self.backend.step(ang, lin, window, tag_index)                    # :197-199
        for k, v in tags.items():                                                # :208-209
            self.tags[k] = v
        pose = np.array(self.backend.pose(), dtype=float)
        self.path.append((pose[0], pose[1]))                                     # :212
        return pose

    def state(self):
        return self.backend.state()

    def close(self):
        self.backend.close()
