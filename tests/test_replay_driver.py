"""Replay driver for the events.csv wire format against the reference's own
replay() loop (tests/golden/replay_events.npz, oracle/gen_golden.py::run_reference_replay).

CPU: parsing, tick latching, windowing and odometry with a CPU backend double built from the oracle.
GPU: the same log through the shipped GpuBackend (EkfSlam on the MI355X)."""
import numpy as np
import pytest
from types import SimpleNamespace

from oracle import ekf_oracle as orc
from tests import golden_util as gu
from tests.conftest import path_ran


class OracleBackend:
    """Test double with the backend protocol of slam_duckietown_amd.replay (CPU, oracle arithmetic)."""

    def __init__(self):
        self.cfg = orc.EkfConfig()
        self.mean, self.cov = np.zeros(3), np.eye(3) * 0.1

    def set_state(self, mean, cov):
        self.mean, self.cov = np.array(mean, dtype=float), np.array(cov, dtype=float)

    def pose(self):
        return self.mean[:3]

    def state(self):
        return self.mean, self.cov

    def step(self, ang, lin, detections, tag_index):
        self.mean, self.cov, tags = orc.ekf_pose_estimation_dense(ang, lin, self.mean, self.cov, 0.7, detections,
                                                                  tag_index, self.cfg)
        return tags

    def close(self):
        pass


def fixture_detector(g):
    names = [str(x) for x in g["frame_names"]]
    table = {n: [] for n in names}
    for fi, tid, t, e in zip(g["det_frame"], g["det_tag_id"], g["det_pose_t"], g["det_err"]):
        table[names[int(fi)]].append(SimpleNamespace(tag_id=int(tid), pose_R=np.eye(3),
                                                     pose_t=np.array(t, dtype=float).reshape(3, 1), pose_err=float(e)))
    seen = {}

    def detect(path, camera_params):
        seen.setdefault("camera_params", list(camera_params))
        import os
        return table[os.path.basename(path)]
    return detect, seen


def check_against_reference_loop(res, g, seen, tol):
    W = len(g["out_size"])
    assert res.windows == W and len(res.poses) == W
    for k in range(W):
        n = int(g["out_size"][k])
        assert np.allclose(res.poses[k], g["out_mean"][k, :3], rtol=0, atol=tol * max(1.0, np.abs(g["out_mean"][k, :3]).max()))
        assert np.allclose(res.path[k + 1], g["out_path"][k], rtol=0, atol=tol)
    n = int(g["out_size"][-1])
    assert orc.rel_fro(res.mean, g["out_mean"][-1, :n]) < tol
    assert orc.rel_fro(res.covariance, g["out_cov"][-1, :n, :n]) < tol
    assert sorted(res.tag_index.items(), key=lambda kv: kv[1]) == [tuple(r) for r in g["out_tag_index"]]
    assert len(res.ground_truth) >= int(g["out_gt_count"][-1])
    assert np.allclose(np.array(res.landmarks, dtype=float), g["out_landmarks"])
    assert seen["camera_params"] == list(g["out_camera_params"])      # K -> [fx, fy, cx, cy] (:181-182)


def test_replay_loop_matches_reference_cpu():
    import slam_duckietown_amd.replay as rp
    g = gu.load("replay_events")
    detect, seen = fixture_detector(g)
    sizes = []
    res = rp.replay(str(g["events_csv"]).splitlines(), backend=OracleBackend(), detector=detect,
                    on_window=lambda d: sizes.append(len(d["tag_index"])))
    check_against_reference_loop(res, g, seen, 1e-11)
    assert sizes == list(g["out_ntags"])


def test_fast_mode_matches_reference_cpu():
    """ENABLE_FAST_MODE (src/replay_no_ros.py:32, :122-123, :216-227): the reference's own replay() with the flag
    set, on the same log (tests/golden/replay_events_fast.npz)."""
    import slam_duckietown_amd.replay as rp
    g, gf = gu.load("replay_events"), gu.load("replay_events_fast")
    detect, seen = fixture_detector(g)
    res = rp.replay(str(g["events_csv"]).splitlines(), backend=OracleBackend(), detector=detect, fast_mode=True)
    merged = dict(g)
    merged.update(gf)
    check_against_reference_loop(res, merged, seen, 1e-11)
    slow = gu.load("replay_events")
    assert not np.array_equal(gf["out_mean"][-1], slow["out_mean"][-1])          # the flag changes the result


def test_detections_event_replaces_images_cpu():
    """The `detections` extension carries the tags in the log itself: same result without a detector."""
    import slam_duckietown_amd.replay as rp
    g = gu.load("replay_events")
    detect, _ = fixture_detector(g)
    lines = []
    for line in str(g["events_csv"]).splitlines():
        stamp, event, data = line.split(",")[:3]
        if event == "image":
            tags = [(t.tag_id, [float(v) for v in t.pose_t.ravel()], t.pose_err) for t in detect(data, [0, 0, 0, 0])]
            line = f"{stamp},detections,{tags!r}"
        lines.append(line)
    a = rp.replay(lines, backend=OracleBackend())
    b = rp.replay(str(g["events_csv"]).splitlines(), backend=OracleBackend(), detector=detect)
    assert np.array_equal(a.mean, b.mean) and np.array_equal(a.covariance, b.covariance) and a.windows == b.windows


def test_unknown_event_and_missing_detector():
    import slam_duckietown_amd.replay as rp
    with pytest.raises(ValueError):
        rp.replay(["1.0,bogus,3"], backend=OracleBackend())
    with pytest.raises(ValueError):
        rp.replay(["1.0,image,frame000000.png"], backend=OracleBackend())


def test_god_mode_presizes_the_state():
    """landmarks event + god_key = ENABLE_GOD_EKF (:140-157): state pre-sized, zero landmark variance."""
    import slam_duckietown_amd.replay as rp
    lines = ["10.0,landmarks,[(1.0, 2.0), (3.0, -1.0)]", "10.1,left_wheel,5", "10.1,right_wheel,5"]
    be = OracleBackend()
    res = rp.replay(lines, backend=be, god_key=[44, 80])
    assert res.tag_index == {44: 0, 80: 1} and res.mean.shape == (7,)
    assert np.array_equal(res.mean[3:], [1.0, 2.0, 3.0, -1.0]) and np.array_equal(res.covariance[3:, 3:], np.zeros((4, 4)))


def god_fixture():
    g, gg = gu.load("replay_events"), gu.load("replay_events_god")
    merged = dict(g)
    merged.update(gg)
    return g, merged, [int(t) for t in gg["god_key"]]


def test_god_mode_matches_reference_cpu():
    """ENABLE_GOD_EKF with GOD_SECRET_KEY = the log's tag ids (src/replay_no_ros.py:23-26, :140-157): the reference's
    own replay() with the flag set, on the same log (tests/golden/replay_events_god.npz)."""
    import slam_duckietown_amd.replay as rp
    g, merged, key = god_fixture()
    detect, seen = fixture_detector(g)
    sizes = []
    res = rp.replay(str(g["events_csv"]).splitlines(), backend=OracleBackend(), detector=detect, god_key=key,
                    on_window=lambda d: sizes.append(len(d["tag_index"])))
    check_against_reference_loop(res, merged, seen, 1e-11)
    assert sizes == list(merged["out_ntags"]) and set(merged["out_size"]) == {3 + 2 * len(key)}
    assert not np.array_equal(merged["out_mean"][-1][:3], g["out_mean"][-1][:3])     # the flag changes the result


@pytest.mark.gpu
@pytest.mark.parametrize("device_association", [False, True])
def test_god_mode_matches_reference_gpu(device_association, both_paths):
    """The god-mode log through the shipped GpuBackend: host association, and the whole front end on the GPU (the
    pre-filled TAG_INDEX goes to the device table)."""
    import slam_duckietown_amd.replay as rp
    g, merged, key = god_fixture()
    detect, seen = fixture_detector(g)
    be = rp.GpuBackend(capacity=3 + 2 * 16, device_association=device_association)
    try:
        res = rp.replay(str(g["events_csv"]).splitlines(), backend=be, detector=detect, god_key=key)
        assert path_ran(be.filt, both_paths)
    finally:
        be.close()
    check_against_reference_loop(res, merged, seen, 1e-9)


@pytest.mark.gpu
def test_replay_loop_matches_reference_gpu(both_paths):
    import slam_duckietown_amd.replay as rp
    g = gu.load("replay_events")
    detect, seen = fixture_detector(g)
    res = rp.replay(str(g["events_csv"]).splitlines(), detector=detect)        # GpuBackend
    check_against_reference_loop(res, g, seen, 1e-9)


@pytest.mark.gpu
def test_fast_mode_matches_reference_gpu(both_paths):
    import slam_duckietown_amd.replay as rp
    g, gf = gu.load("replay_events"), gu.load("replay_events_fast")
    detect, seen = fixture_detector(g)
    res = rp.replay(str(g["events_csv"]).splitlines(), detector=detect, fast_mode=True)     # GpuBackend
    merged = dict(g)
    merged.update(gf)
    check_against_reference_loop(res, merged, seen, 1e-9)


@pytest.mark.gpu
def test_replay_loop_with_device_association_gpu(both_paths):
    """The same log with the whole front end on the GPU (GpuBackend(device_association=True))."""
    import slam_duckietown_amd.replay as rp
    g = gu.load("replay_events")
    detect, seen = fixture_detector(g)
    be = rp.GpuBackend(capacity=3 + 2 * 16, device_association=True)
    try:
        res = rp.replay(str(g["events_csv"]).splitlines(), backend=be, detector=detect)
        assert path_ran(be.filt, both_paths)
    finally:
        be.close()
    check_against_reference_loop(res, g, seen, 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("device_association", [False, True])
def test_replay_loop_on_the_small_state_path_gpu(device_association, monkeypatch):
    """The reference's loop at its real map size on the path it takes by default: GpuBackend's first handle holds 38
    landmarks, so every window is ONE launch of the small-state kernel, and with the host association the state comes back
    with the step (ekf_step_fetch: pose() and state() answer from what it brought) -- launches and fetches counted.  Same
    fixture -- the reference's own replay() -- same bar."""
    import slam_duckietown_amd as sd
    import slam_duckietown_amd.replay as rp
    monkeypatch.setenv("EKFSLAM_HIP_SMALL_STATE", "1")
    g = gu.load("replay_events")
    detect, seen = fixture_detector(g)
    be = rp.GpuBackend(device_association=device_association)
    try:
        assert be.filt.n_max == 79
        res = rp.replay(str(g["events_csv"]).splitlines(), backend=be, detector=detect)
        lib = sd.load_library()
        W = len(g["out_size"])
        assert lib.ekf_debug_small_launches(be.filt._h) >= W
        assert lib.ekf_debug_fused_fetches(be.filt._h) == (0 if device_association else W)
        mean, cov = be.state()                               # (what the last step brought back) == the device's own copy
        dm, dc = be.filt.state()
        assert np.array_equal(mean, dm) and np.array_equal(cov, dc)
    finally:
        be.close()
    check_against_reference_loop(res, g, seen, 1e-9)
