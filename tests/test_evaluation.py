"""Accuracy / consistency evaluation."""
import math

import numpy as np
import pytest

from oracle import ekf_oracle as orc


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
def test_block_download_and_monte_carlo_nees():
    """Pose NEES over a Monte-Carlo bank of 24 trajectories (different noise and maps per trajectory).
    With perfect odometry in the stream and the reference's generous noise constants the filter must
    come out conservative: average NEES below the upper chi-square bound."""
    import slam_duckietown_amd as sd
    import slam_duckietown_amd.evaluation as ev
    import slam_duckietown_amd.synthetic as syn
    N, steps, B = 40, 30, 24
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
