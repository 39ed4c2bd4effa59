"""Helpers to turn the committed golden arrays back into the reference's input objects."""
import os
from types import SimpleNamespace

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def detections_for_step(g, k):
    """[(timestamp, [tag...])] for window k, frames in order, tags in recorded order."""
    sel = np.nonzero(g["det_step"] == k)[0]
    frames = {}
    for i in sel:
        frames.setdefault(int(g["det_frame"][i]), []).append(
            SimpleNamespace(tag_id=int(g["det_tag_id"][i]), pose_R=np.eye(3),
                            pose_t=g["det_pose_t"][i].reshape(3, 1).copy(), pose_err=float(g["det_err"][i])))
    return [(float(k) + 0.1 * f, tags) for f, tags in sorted(frames.items())]


def tags_from_obs(idx, zr, zb):
    """Synthetic-stream observations as AprilTag-like objects (pose_t = [[-y_r],[0],[x_r]])."""
    out = []
    for i, r, b in zip(idx, zr, zb):
        xr, yr = r * np.cos(b), r * np.sin(b)
        out.append(SimpleNamespace(tag_id=1000 + int(i), pose_R=np.eye(3),
                                   pose_t=np.array([[-yr], [0.0], [xr]]), pose_err=0.0))
    return out


REPLAY_CASES = ["replay_default", "replay_bigturns", "replay_no_measurement", "replay_no_motion",
                "replay_linear_interp", "replay_ignore_tags"]


def ignore_tags(g):
    """IGNORE_TAGS the fixture was generated with (src/replay_no_ros.py:36-37, :286); () for the default list."""
    return tuple(int(t) for t in g["ignore_tags"]) if "ignore_tags" in g else ()
