"""CPU: the oracle (oracle/ekf_oracle.py) against the golden vectors produced by the reference itself.

These pin the oracle; the GPU parity tests then compare the HIP path with the pinned oracle.
"""
import numpy as np
import pytest

from oracle import ekf_oracle as orc
from tests import golden_util as gu

TOL = 1e-12      # dense restatement vs reference: same BLAS calls, expect ~1e-15
TOL_S = 1e-9     # structured O(n^2) formulation vs reference (bar for the product is 1e-6)


def cfg_of(g):
    return orc.EkfConfig(enable_measurement_model=bool(g["flag_measurement"]),
                         enable_circular_interpolation=bool(g["flag_circular"]),
                         disable_motion_model=bool(g["flag_no_motion"]), ignore_tags=gu.ignore_tags(g))


@pytest.mark.parametrize("case", gu.REPLAY_CASES)
def test_whole_function_dense(case):
    g = gu.load(case)
    cfg = cfg_of(g)
    mean = np.zeros(3)
    cov = np.eye(3) * 0.1
    tag_index = {}
    for k in range(len(g["lin"])):
        det = gu.detections_for_step(g, k)
        mean, cov, tp = orc.ekf_pose_estimation_dense(g["ang"][k], g["lin"][k], mean, cov, 0.7, det, tag_index, cfg)
        n = int(g["out_size"][k])
        assert len(mean) == n
        order = [i for i in g["out_obs_order"][k] if i >= 0]
        assert list(tp.keys()) == order
        assert orc.rel_fro(mean, g["out_mean"][k, :n]) < TOL
        assert orc.rel_fro(cov, g["out_cov"][k, :n, :n]) < TOL
    assert sorted(tag_index.items(), key=lambda kv: kv[1]) == [tuple(r) for r in g["out_tag_index"]]


@pytest.mark.parametrize("case", gu.REPLAY_CASES)
def test_whole_function_structured(case):
    """association/augment on host + the structured step = what the GPU path does."""
    g = gu.load(case)
    cfg = cfg_of(g)
    mean = np.zeros(3)
    cov = np.eye(3) * 0.1
    tag_index = {}
    for k in range(len(g["lin"])):
        det = gu.detections_for_step(g, k)
        tp = orc.associate(det, tag_index, mean, cfg)
        mean, cov = orc.augment(mean, cov, len(tag_index), tp, cfg)
        idx = list(tp.keys())
        mean, cov = orc.ekf_step_structured(mean, cov, g["lin"][k], g["ang"][k], idx,
                                            [tp[i][4] for i in idx], [tp[i][5] for i in idx], cfg)
        n = int(g["out_size"][k])
        assert orc.rel_fro(mean, g["out_mean"][k, :n]) < TOL_S
        assert orc.rel_fro(cov, g["out_cov"][k, :n, :n]) < TOL_S


@pytest.mark.parametrize("case,step_fn,tol", [
    ("stream_n20_m8", orc.ekf_step_dense, TOL), ("stream_n20_m8", orc.ekf_step_structured, TOL_S),
    ("stream_n20_m1", orc.ekf_step_dense, TOL), ("stream_n20_m1", orc.ekf_step_structured, TOL_S),
    ("stream_n50_m8", orc.ekf_step_dense, TOL), ("stream_n50_m8", orc.ekf_step_structured, TOL_S),
])
def test_stream_small(case, step_fn, tol):
    g = gu.load(case)
    cfg = orc.EkfConfig()
    mean = g["mean0"].copy()
    cov = np.diag(g["diag0"])
    kept = {int(s): i for i, s in enumerate(g["out_cov_steps"])}
    for k in range(len(g["lin"])):
        mean, cov = step_fn(mean, cov, g["lin"][k], g["ang"][k], g["idx"][k], g["zr"][k], g["zb"][k], cfg)
        assert orc.rel_fro(mean, g["out_mean"][k]) < tol, k
        assert orc.rel_fro(np.diag(cov), g["out_diag"][k]) < tol, k
        assert abs(np.linalg.norm(cov) - g["out_fro"][k]) < tol * g["out_fro"][k], k
        if k in kept:
            assert orc.rel_fro(cov, g["out_cov"][kept[k]]) < tol, k


@pytest.mark.parametrize("step_fn,tol", [(orc.ekf_step_dense, TOL), (orc.ekf_step_structured, TOL_S)])
def test_stream_n500(step_fn, tol):
    g = gu.load("stream_n500_m8")
    cfg = orc.EkfConfig()
    mean = g["mean0"].copy()
    cov = np.diag(g["diag0"])
    kept = {int(s): i for i, s in enumerate(g["out_cov_steps"])}
    rows = g["out_cov_rows"]
    for k in range(len(g["lin"])):
        mean, cov = step_fn(mean, cov, g["lin"][k], g["ang"][k], g["idx"][k], g["zr"][k], g["zb"][k], cfg)
        assert orc.rel_fro(mean, g["out_mean"][k]) < tol
        assert orc.rel_fro(np.diag(cov), g["out_diag"][k]) < tol
        if k in kept:
            assert orc.rel_fro(cov[rows, :], g["out_cov"][kept[k]]) < tol
    assert orc.rel_fro(cov.sum(axis=1), g["out_cov_rowsum"]) < 1e-9
    assert orc.rel_fro(cov.sum(axis=0), g["out_cov_colsum"]) < 1e-9


def test_synthetic_stream_is_the_fixture_input():
    """The committed stream inputs are what synthetic_stream() regenerates (same seeds)."""
    g = gu.load("stream_n50_m8")
    mean0, diag0, lin, ang, idx, zr, zb = orc.synthetic_stream(50, 30, 8, 0)
    assert np.array_equal(mean0, g["mean0"]) and np.array_equal(diag0, g["diag0"])
    assert np.array_equal(idx, g["idx"]) and np.array_equal(lin, g["lin"]) and np.array_equal(ang, g["ang"])
    # the fixture stores range/bearing after the reference's pose_t round trip: equal to rounding
    assert np.allclose(zr, g["zr"], rtol=1e-14, atol=0) and np.allclose(zb, g["zb"], rtol=1e-13, atol=1e-15)


def test_proto3():
    g = gu.load("proto3")
    state = np.array([0.0, 0.0, 0.0])
    cov = np.eye(3) * 0.1
    for k in range(len(g["dt"])):
        state, cov = orc.predict3(state, cov, tuple(g["control"][k]), g["dt"][k])
        assert np.allclose(state, g["pred_state"][k], rtol=1e-13, atol=1e-15)
        assert np.allclose(cov, g["pred_cov"][k], rtol=1e-13, atol=1e-15)
        state, cov = orc.update3(state, cov, tuple(g["observation"][k]), g["landmark"][k])
        assert np.allclose(state, g["upd_state"][k], rtol=1e-13, atol=1e-15)
        assert np.allclose(cov, g["upd_cov"][k], rtol=1e-13, atol=1e-15)


def test_odometry():
    g = gu.load("odometry")
    for (a, b, c, d), dphi, disp in zip(g["ticks"], g["dphi"], g["disp"]):
        l = orc.delta_phi(int(a), int(b), int(g["resolution"]))
        r = orc.delta_phi(int(c), int(d), int(g["resolution"]))
        assert (l, r) == tuple(dphi)
        assert orc.displacement(float(g["wheel_radius"]), float(g["baseline"]), l, r) == tuple(disp)


# ---------------------------------------------------------------------------------------------
# the algebra the HIP kernels run (upper triangle + deferred ranks), restated in NumPy
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case,rank_limit", [("stream_n20_m8", 64), ("stream_n20_m8", 16), ("stream_n20_m1", 64),
                                             ("stream_n50_m8", 48)])
def test_deferred_symmetric_restatement_golden(case, rank_limit):
    g = gu.load(case)
    f = orc.DeferredSymmetricFilter(g["mean0"], g["diag0"], orc.EkfConfig(), rank_limit)
    kept = {int(s): i for i, s in enumerate(g["out_cov_steps"])}
    for k in range(len(g["lin"])):
        f.step(g["lin"][k], g["ang"][k], g["idx"][k], g["zr"][k], g["zb"][k])
        assert orc.rel_fro(f.mean, g["out_mean"][k]) < TOL_S
        if k in kept:
            assert orc.rel_fro(f.covariance(), g["out_cov"][kept[k]]) < TOL_S


def test_deferred_symmetric_long_run_stays_on_the_dense_path():
    """1500 steps: no drift between the symmetric deferred form and the reference-shaped dense update."""
    N, steps, m = 20, 1500, 8
    mean0, diag0, lin, ang, idx, zr, zb = orc.synthetic_stream(N, steps, m, 2)
    cfg = orc.EkfConfig()
    f = orc.DeferredSymmetricFilter(mean0, diag0, cfg)
    om, oP = mean0.copy(), np.diag(diag0)
    for k in range(steps):
        om, oP = orc.ekf_step_dense(om, oP, lin[k], ang[k], idx[k], zr[k], zb[k], cfg)
        f.step(lin[k], ang[k], idx[k], zr[k], zb[k])
        if k % 250 == 249:
            assert orc.rel_fro(f.mean, om) < 1e-10
            assert orc.rel_fro(f.covariance(), oP) < 1e-10
