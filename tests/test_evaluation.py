"""Accuracy / consistency evaluation."""
import math

import numpy as np
import pytest

from oracle import ekf_oracle as orc

from tests.conftest import path_ran


def test_vicon_alignment_matches_reference_formula():
    """align_vicon == the per-point rotate_around + normalise of scripts/decode_bag_file.py:176-178."""
    import slam_duckietown_amd.evaluation as ev
    rng = np.random.default_rng(1)
    pts = rng.uniform(-3000, 3000, (50, 2))
    ix, iy, ith = 812.5, -440.25, 0.83
    ref = []
    for x, y in pts:                         # the reference's own arithmetic, point by point
        rx = ix + (x - ix) * math.cos(ith) - (y - iy) * math.sin(ith)
        ry = iy + (x - ix) * math.sin(ith) + (y - iy) * math.cos(ith)
        ref.append(((rx - ix) / 1000, (ry - iy) / 1000))
    assert np.allclose(ev.align_vicon(pts, ix, iy, ith), np.array(ref), rtol=1e-14, atol=1e-15)
    assert np.allclose(ev.rotate_around(1.0, 2.0, 3.0, 5.0, 0.4),
                       (1 + 2 * math.cos(0.4) - 3 * math.sin(0.4), 2 + 2 * math.sin(0.4) + 3 * math.cos(0.4)))


def test_vicon_ground_truth_matches_the_reference_decoder_on_the_recorded_data():
    """`vicon_ground_truth` against the reference's own `get_ground_truth` (scripts/decode_bag_file.py:111-253)
    executed on bags/quackgpt_small_town_joystick{.csv,_trajectories.csv,.xcp}: every ground-truth time and
    position and every landmark, bit for bit (tests/golden/vicon_alignment.npz, oracle/gen_golden.py)."""
    import slam_duckietown_amd.evaluation as ev
    from tests import golden_util as gu
    from tests import vicon_scenario as vs
    g = gu.load("vicon_alignment")
    r = ev.vicon_ground_truth(vs.decode(g["robot_fields"]), vs.decode(g["marker_fields"]), float(g["start_capture_time"]),
                              float(g["end_capture_time"]), float(g["first_timestamp"]), float(g["delay"]),
                              g["robot_len"], g["marker_len"])
    assert r["framerate"] == 100 and len(r["times"]) == 4846
    assert np.array_equal(r["times"], g["out_gt_time"])
    assert np.array_equal(r["xy"], g["out_gt_xy"])
    assert np.array_equal(r["landmarks"], g["out_landmarks"]) and r["landmarks"].shape == (13, 2)
    assert r["landmarks_time"] == float(g["out_landmarks_time"])
    # the per-point helpers agree with the same outputs
    x0, y0, th0 = r["origin"]
    pts = vs.decode(g["robot_fields"])[:, 5:7]
    assert np.allclose(ev.align_vicon(pts, x0, y0, th0), g["out_gt_xy"], rtol=0, atol=1e-15)
    rx, ry = ev.rotate_around(x0, y0, pts[17, 0], pts[17, 1], th0)
    assert ((rx - x0) / 1000, (ry - y0) / 1000) == tuple(g["out_gt_xy"][17])
    # a recording that does not overlap the bag is an error (the reference prints and exits, :157-164)
    with pytest.raises(ValueError):
        ev.vicon_ground_truth(vs.decode(g["robot_fields"]), vs.decode(g["marker_fields"]), float(g["start_capture_time"]),
                              float(g["end_capture_time"]), float(g["first_timestamp"]) + 500.0)


def test_vicon_csv_loader_reads_the_reference_files():
    """The loader on the reference's own files gives the fixture's fields (only where /root/reference exists)."""
    import os
    import slam_duckietown_amd.evaluation as ev
    from tests import golden_util as gu
    from tests import vicon_scenario as vs
    prefix = "/root/reference/bags/quackgpt_small_town_joystick"
    if not os.path.exists(prefix + ".csv"):
        pytest.skip("reference data not present (GPU box)")
    g = gu.load("vicon_alignment")
    for path, key in ((prefix + ".csv", "robot"), (prefix + "_trajectories.csv", "marker")):
        fields, lens = ev.load_vicon_csv(path)
        assert np.array_equal(lens, g[key + "_len"])
        assert np.array_equal(fields, vs.decode(g[key + "_fields"]), equal_nan=True)


def test_vicon_track_scenario_oracle_ate():
    """The filter (oracle arithmetic) follows the recorded Vicon track to a few centimetres and maps its 13 markers."""
    import slam_duckietown_amd.evaluation as ev
    from tests import vicon_scenario as vs
    xy, wins = vs.windows(0)
    mu, P, ti = np.zeros(3), np.eye(3) * 0.1, {}
    path = [mu[:2].copy()]
    for ang, lin, det in wins:
        mu, P, _ = orc.ekf_pose_estimation_dense(ang, lin, mu, P, 0.7, det, ti, orc.EkfConfig())
        path.append(mu[:2].copy())
    assert len(wins) == 67 and len(ti) == 13
    assert ev.ate_rmse(path, xy) < 0.06 and ev.ate_rmse(path, xy, align=True) < 0.04


def test_ate_and_rigid_alignment():
    import slam_duckietown_amd.evaluation as ev
    rng = np.random.default_rng(2)
    truth = np.cumsum(rng.normal(0, 0.1, (200, 2)), axis=0)
    assert ev.ate_rmse(truth, truth) == 0.0
    shifted = truth + np.array([0.3, -0.4])
    assert abs(ev.ate_rmse(shifted, truth) - 0.5) < 1e-12
    th = 0.7
    R = np.array([[math.cos(th), -math.sin(th)], [math.sin(th), math.cos(th)]])
    moved = truth @ R.T + np.array([2.0, 1.0])
    assert ev.ate_rmse(moved, truth, align=True) < 1e-12
    t = np.linspace(0, 10, 200)
    assert np.allclose(ev.resample_truth(t, truth, t[::7]), truth[::7])


def test_nees_statistics():
    import slam_duckietown_amd.evaluation as ev
    rng = np.random.default_rng(3)
    B, d = 400, 3
    A = rng.normal(size=(d, d))
    P = A @ A.T + np.eye(d)
    e = rng.multivariate_normal(np.zeros(d), P, size=B)
    vals = ev.nees(e, np.broadcast_to(P, (B, d, d)))
    lo, hi = ev.chi2_bounds(d, B)
    assert lo < vals.mean() < hi            # a consistent sample passes the ANEES test
    assert not (lo < ev.nees(3 * e, np.broadcast_to(P, (B, d, d))).mean() < hi)   # over-confident fails


@pytest.mark.gpu
def test_block_download_and_monte_carlo_nees(both_paths):
    """Pose NEES over a Monte-Carlo bank of 24 trajectories (different noise and maps per trajectory).
    With perfect odometry in the stream and the reference's generous noise constants the filter must
    come out conservative: average NEES below the upper chi-square bound."""
    import slam_duckietown_amd as sd
    import slam_duckietown_amd.evaluation as ev
    import slam_duckietown_amd.synthetic as syn
    N, steps, B = 36, 30, 24            # (36 landmarks: n = 75, the small-state path by default)
    streams = [syn.synthetic_stream(N, steps, 8, t) for t in range(B)]
    with sd.EkfSlam(3 + 2 * N, batch=B) as f:
        for b, s in enumerate(streams):
            f.set_state_diag(s[0], s[1], b)
        f.run_stream(np.stack([s[2] for s in streams], 1), np.stack([s[3] for s in streams], 1),
                     np.stack([s[4] for s in streams], 1), np.stack([s[5] for s in streams], 1),
                     np.stack([s[6] for s in streams], 1))
        P5 = f.covariance(5)
        assert np.array_equal(f.covariance_block(0, 0, 3, 3, 5), P5[:3, :3])
        assert np.array_equal(f.covariance_block(7, 2, 4, 9, 5), P5[7:11, 2:11])
        truth = syn.true_poses(steps)[-1]
        vals, avg, (lo, hi) = ev.pose_nees(f, np.tile(truth, (B, 1)))
        assert vals.shape == (B,) and (vals >= 0).all()
        assert avg < hi
        path_err = ev.ate_rmse([f.mean(b)[:2] for b in range(B)], np.tile(truth[:2], (B, 1)))
        assert path_err < 0.05
        assert path_ran(f, both_paths)


@pytest.mark.gpu
def test_monte_carlo_nees_and_ate_equal_the_oracle_bank(both_paths):
    """The same Monte-Carlo bank through the HIP path and through the oracle: per-trajectory NEES, their average and
    the ATE are the oracle's numbers (not merely inside a bound)."""
    import slam_duckietown_amd as sd
    import slam_duckietown_amd.evaluation as ev
    import slam_duckietown_amd.synthetic as syn
    N, steps, B = 36, 30, 16            # (n = 75: the small-state path by default)
    streams = [syn.synthetic_stream(N, steps, 8, 100 + t) for t in range(B)]
    truth = syn.true_poses(steps)[-1]
    cfg = orc.EkfConfig()
    o_err, o_cov, o_xy = [], [], []
    for s in streams:
        om, oP = s[0].copy(), np.diag(s[1])
        for k in range(steps):
            om, oP = orc.ekf_step_dense(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], cfg)
        e = om[:3] - truth
        e[2] = ev.wrap_angle(e[2])
        o_err.append(e)
        o_cov.append(oP[:3, :3])
        o_xy.append(om[:2])
    o_nees = ev.nees(np.array(o_err), np.array(o_cov))
    with sd.EkfSlam(3 + 2 * N, batch=B) as f:
        for b, s in enumerate(streams):
            f.set_state_diag(s[0], s[1], b)
        f.run_stream(np.stack([s[2] for s in streams], 1), np.stack([s[3] for s in streams], 1),
                     np.stack([s[4] for s in streams], 1), np.stack([s[5] for s in streams], 1),
                     np.stack([s[6] for s in streams], 1))
        vals, avg, (lo, hi) = ev.pose_nees(f, np.tile(truth, (B, 1)))
        g_xy = [f.mean(b)[:2] for b in range(B)]
        assert path_ran(f, both_paths)
    assert np.allclose(vals, o_nees, rtol=1e-7, atol=1e-12)
    assert abs(avg - o_nees.mean()) <= 1e-7 * o_nees.mean()
    ate_g, ate_o = ev.ate_rmse(g_xy, np.tile(truth[:2], (B, 1))), ev.ate_rmse(o_xy, np.tile(truth[:2], (B, 1)))
    assert abs(ate_g - ate_o) <= 1e-9 * max(ate_o, 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("device_association", [False, True])
def test_vicon_track_ate_on_the_gpu(device_association, both_paths):
    """The recorded Vicon track through the shipped GpuBackend (host and device front end): state equal to the
    oracle's on the same windows, ATE against the Vicon truth equal to the oracle's and within a few centimetres."""
    import slam_duckietown_amd.evaluation as ev
    from slam_duckietown_amd.replay import GpuBackend
    from tests import vicon_scenario as vs
    xy, wins = vs.windows(0)
    om, oP, oti = np.zeros(3), np.eye(3) * 0.1, {}
    opath = [om[:2].copy()]
    for ang, lin, det in wins:
        om, oP, _ = orc.ekf_pose_estimation_dense(ang, lin, om, oP, 0.7, det, oti, orc.EkfConfig())
        opath.append(om[:2].copy())
    be = GpuBackend(capacity=3 + 2 * 16, device_association=device_association)
    try:
        be.set_state(np.zeros(3), np.eye(3) * 0.1)
        ti, path = {}, [np.zeros(2)]
        for ang, lin, det in wins:
            be.step(ang, lin, det, ti)
            path.append(np.array(be.pose()[:2]))
        mu, P = be.state()
        assert path_ran(be.filt, both_paths)
    finally:
        be.close()
    assert ti == oti
    assert orc.rel_fro(mu, om) < 1e-9 and orc.rel_fro(P, oP) < 1e-9
    ate_g, ate_o = ev.ate_rmse(path, xy), ev.ate_rmse(opath, xy)
    assert abs(ate_g - ate_o) < 1e-9 and ate_g < 0.06
