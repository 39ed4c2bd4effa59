"""ROS-free node adapter: callback sequencing of histogram_lane_filter_node.py:148-217."""
import numpy as np
import pytest
from types import SimpleNamespace

from oracle import ekf_oracle as orc
from tests.test_replay_driver import OracleBackend


def script(seed=3, ticks=60):
    """A message sequence: encoder messages, frames with tags, timer ticks (some with no encoder motion)."""
    rng = np.random.default_rng(seed)
    lm = {7: (0.8, 0.2), 12: (0.6, -0.3), 31: (1.0, 0.5)}
    msgs, lt, rt = [], 0, 0
    for k in range(ticks):
        if k % 5 != 4:
            lt += int(rng.integers(0, 6)); rt += int(rng.integers(0, 6))
            msgs.append(("L", lt)); msgs.append(("R", rt))
        if k % 2 == 0:
            tags = [SimpleNamespace(tag_id=i, pose_R=np.eye(3), pose_err=1e-3,
                                    pose_t=np.array([[-y + rng.normal(0, 0.01)], [0.0], [x + rng.normal(0, 0.01)]]))
                    for i, (x, y) in lm.items() if rng.random() < 0.7]
            msgs.append(("I", k, tags))
        msgs.append(("T",))
    return msgs


def restated_node(msgs, persistent):
    """The node's own bookkeeping (:148-217), restated with the oracle as the EKF."""
    cfg = orc.EkfConfig()
    mu, Sigma = np.array([0.0, 0.0, 0.0]), np.eye(3) * 0.1
    left = right = dl = dr = 0
    det, poses, TI = [], [], {}
    for msg in msgs:
        if msg[0] == "L":
            dl = msg[1] - left
        elif msg[0] == "R":
            dr = msg[1] - right
        elif msg[0] == "I":
            if msg[2]:
                det.append((msg[1], msg[2]))
        else:
            if dl == 0 and dr == 0:
                poses.append(None)
                continue
            left += dl; right += dr
            a, l = orc.displacement(0.0318, 0.1, orc.delta_phi(dl, 0, 135), orc.delta_phi(dr, 0, 135))
            dl = dr = 0
            mu, Sigma, _ = orc.ekf_pose_estimation_dense(a, l, mu, Sigma, 0.0, det, TI if persistent else {}, cfg)
            det = []
            poses.append(mu[:3].copy())
    return poses, mu, Sigma


def drive(adapter, msgs):
    poses = []
    for msg in msgs:
        if msg[0] == "L":
            adapter.on_left_encoder(msg[1], 135)
        elif msg[0] == "R":
            adapter.on_right_encoder(msg[1], 135)
        elif msg[0] == "I":
            adapter.on_image(msg[1], msg[2])
        else:
            poses.append(adapter.on_timer())
    return poses


@pytest.mark.parametrize("persistent", [True, False])
def test_adapter_sequencing_cpu(persistent):
    from slam_duckietown_amd.node_adapter import EkfNodeAdapter
    msgs = script()
    ref_poses, ref_mu, ref_S = restated_node(msgs, persistent)
    ad = EkfNodeAdapter(backend=OracleBackend(), persistent_tag_index=persistent)
    poses = drive(ad, msgs)
    assert len(poses) == len(ref_poses)
    for p, r in zip(poses, ref_poses):
        assert (p is None) == (r is None)
        if p is not None:
            assert np.allclose(p, r, rtol=1e-12, atol=1e-14)
    mu, S = ad.state()
    assert orc.rel_fro(mu, ref_mu) < 1e-12 and orc.rel_fro(S, ref_S) < 1e-12
    if persistent:
        assert ad.tag_index == {t: i for i, t in enumerate(sorted(ad.tag_index, key=ad.tag_index.get))}


@pytest.mark.gpu
@pytest.mark.parametrize("device_association", [False, True])
def test_adapter_on_gpu(device_association, both_paths):
    from slam_duckietown_amd.node_adapter import EkfNodeAdapter
    msgs = script(seed=5)
    ref_poses, ref_mu, ref_S = restated_node(msgs, True)
    ad = EkfNodeAdapter(device_association=device_association)
    try:
        poses = drive(ad, msgs)
        mu, S = ad.state()
        from tests.conftest import path_ran
        assert path_ran(ad.backend.filt, both_paths)
    finally:
        ad.close()
    for p, r in zip(poses, ref_poses):
        assert (p is None) == (r is None)
        if p is not None:
            assert np.allclose(p, r, rtol=1e-9, atol=1e-11)
    assert orc.rel_fro(mu, ref_mu) < 1e-9 and orc.rel_fro(S, ref_S) < 1e-9
