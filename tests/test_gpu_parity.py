"""GPU parity: the HIP path (through the C ABI) against the pinned oracle and the golden vectors.

Bar (BASELINE.json north_star): state and covariance within 1e-6 relative Frobenius norm of the
reference NumPy path on identical inputs.  REL_TOL is that bar; TIGHT is what fp64 kernels are
expected to reach and is asserted too so that a regression shows long before the bar.
"""
import numpy as np
import pytest

from oracle import ekf_oracle as orc
from tests import golden_util as gu
from tests.conftest import path_ran

pytestmark = pytest.mark.gpu

REL_TOL = 1e-6
TIGHT = 1e-9


@pytest.fixture(scope="module")
def sd():
    import slam_duckietown_amd as sd
    sd.load_library()          # fails loudly if the HIP library is not built
    return sd


def close(a, b, tol=TIGHT):
    r = orc.rel_fro(a, b)
    assert r < REL_TOL, f"rel Frobenius {r:.3e} exceeds the 1e-6 bar"
    assert r < tol, f"rel Frobenius {r:.3e} exceeds the expected {tol:g}"


# ---------------------------------------------------------------------------------------------
# golden vectors produced by the reference itself
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["stream_n20_m8", "stream_n20_m1", "stream_n50_m8"])
def test_stream_golden_every_step(sd, case, both_paths):
    g = gu.load(case)
    n = len(g["mean0"])
    kept = {int(s): i for i, s in enumerate(g["out_cov_steps"])}
    with sd.EkfSlam(n) as f:
        f.set_state_diag(g["mean0"], g["diag0"])
        for k in range(len(g["lin"])):
            f.step(g["lin"][k], g["ang"][k], g["idx"][k], g["zr"][k], g["zb"][k])
            close(f.mean(), g["out_mean"][k])
            if k in kept:
                close(f.covariance(), g["out_cov"][kept[k]])
        assert f.flags() == 0 and path_ran(f, both_paths)


def test_stream_golden_n500(sd):
    g = gu.load("stream_n500_m8")
    n = len(g["mean0"])
    kept = {int(s): i for i, s in enumerate(g["out_cov_steps"])}
    rows = g["out_cov_rows"]
    with sd.EkfSlam(n) as f:
        f.set_state_diag(g["mean0"], g["diag0"])
        for k in range(len(g["lin"])):
            f.step(g["lin"][k], g["ang"][k], g["idx"][k], g["zr"][k], g["zb"][k])
            mu, P = f.state()
            close(mu, g["out_mean"][k])
            close(np.diag(P), g["out_diag"][k])
            assert abs(np.linalg.norm(P) - g["out_fro"][k]) < TIGHT * g["out_fro"][k]
            if k in kept:
                close(P[rows, :], g["out_cov"][kept[k]])
        close(P.sum(axis=1), g["out_cov_rowsum"])
        close(P.sum(axis=0), g["out_cov_colsum"])


@pytest.mark.parametrize("case", gu.REPLAY_CASES)
def test_drop_in_function_golden(sd, case, both_paths):
    """EKF_pose_estimation drop-in: association, gate, averaging, augmentation, flags, wraps."""
    from slam_duckietown_amd import ekf_bindings as eb
    g = gu.load(case)
    eb.DROP_IN_CONFIG = sd.EkfConfig(enable_measurement_model=bool(g["flag_measurement"]),
                                     enable_circular_interpolation=bool(g["flag_circular"]),
                                     disable_motion_model=bool(g["flag_no_motion"]), ignore_tags=gu.ignore_tags(g))
    try:
        mean = np.array([0.0, 0.0, 0.0])
        cov = np.eye(3) * 0.1
        tag_index = {}
        for k in range(len(g["lin"])):
            det = gu.detections_for_step(g, k)
            mean, cov, tp = eb.EKF_pose_estimation(g["ang"][k], g["lin"][k], mean, cov, 0.7, det, tag_index)
            n = int(g["out_size"][k])
            assert mean.shape == (n,) and cov.shape == (n, n)
            assert list(tp.keys()) == [i for i in g["out_obs_order"][k] if i >= 0]
            close(mean, g["out_mean"][k, :n])
            close(cov, g["out_cov"][k, :n, :n])
        assert sorted(tag_index.items(), key=lambda kv: kv[1]) == [tuple(r) for r in g["out_tag_index"]]
        assert path_ran(eb._drop.filt, both_paths)
    finally:
        eb.DROP_IN_CONFIG = sd.EkfConfig()


def test_drop_in_fresh_arrays_are_uploaded(sd, both_paths):
    """A caller that passes arrays we did not return (or edits them) must not hit the resident state."""
    from slam_duckietown_amd import ekf_bindings as eb
    g = gu.load("replay_default")
    mean = np.array([0.0, 0.0, 0.0])
    cov = np.eye(3) * 0.1
    ti = {}
    ocfg = orc.EkfConfig()
    omean, ocov, oti = mean.copy(), cov.copy(), {}
    for k in range(12):
        det = gu.detections_for_step(g, k)
        mean, cov, _ = eb.EKF_pose_estimation(g["ang"][k], g["lin"][k], mean, cov, 0.7, det, ti)
        omean, ocov, _ = orc.ekf_pose_estimation_dense(g["ang"][k], g["lin"][k], omean, ocov, 0.7, det, oti, ocfg)
        if k % 3 == 0:             # perturb in place / rebind to copies: the wrapper must notice
            mean = mean.copy()
            cov = cov * 1.0
            cov[0, 0] *= 1.5
            ocov[0, 0] *= 1.5
        close(mean, omean)
        close(cov, ocov)
    assert path_ran(eb._drop.filt, both_paths)


def test_drop_in_keyerror_like_reference(sd, both_paths):
    """A TAG_INDEX larger than the state with no measurement for the new index: KeyError (:359)."""
    from types import SimpleNamespace
    from slam_duckietown_amd import ekf_bindings as eb
    tag = SimpleNamespace(tag_id=5, pose_R=np.eye(3), pose_t=np.array([[0.1], [0.0], [0.7]]), pose_err=0.0)
    with pytest.raises(KeyError):
        eb.EKF_pose_estimation(0.0, 0.01, np.zeros(3), np.eye(3) * 0.1, 0.7, [(0.0, [tag])], {99: 0})


# ---------------------------------------------------------------------------------------------
# oracle on seeded inputs
# ---------------------------------------------------------------------------------------------
def test_config2_n500_fifty_steps_vs_dense(sd):
    """BASELINE config 2: N=500, 1 trajectory, every step compared for 50 steps."""
    N, steps, m = 500, 50, 8
    mean0, diag0, lin, ang, idx, zr, zb = orc.synthetic_stream(N, steps, m, 0)
    cfg = orc.EkfConfig()
    om, oP = mean0.copy(), np.diag(diag0)
    with sd.EkfSlam(len(mean0)) as f:
        f.set_state_diag(mean0, diag0)
        for k in range(steps):
            f.step(lin[k], ang[k], idx[k], zr[k], zb[k])
            om, oP = orc.ekf_step_dense(om, oP, lin[k], ang[k], idx[k], zr[k], zb[k], cfg)
            mu, P = f.state()
            close(mu, om)
            close(P, oP)


def test_config3_n2000_stream_vs_structured(sd):
    """BASELINE config 3 size (n = 4003): 12 steps through run_stream vs the O(n^2) oracle, plus
    size-independent properties (symmetry to rounding, variances never grow under updates)."""
    N, steps, m = 2000, 12, 8
    mean0, diag0, lin, ang, idx, zr, zb = orc.synthetic_stream(N, steps, m, 3)
    cfg = orc.EkfConfig()
    om, oP = mean0.copy(), np.diag(diag0)
    for k in range(steps):
        om, oP = orc.ekf_step_structured(om, oP, lin[k], ang[k], idx[k], zr[k], zb[k], cfg)
    with sd.EkfSlam(len(mean0)) as f:
        f.set_state_diag(mean0, diag0)
        f.run_stream(lin, ang, idx, zr, zb)
        mu, P = f.state()
    close(mu, om)
    close(P, oP)
    assert np.abs(P - P.T).max() <= 1e-9 * np.abs(P).max()
    d = np.diag(P)
    assert (d[3:] <= diag0[3:] * (1 + 1e-12)).all() and (d > 0).all()


def test_one_step_n2000_vs_dense(sd):
    """One full-size step against the reference-shaped dense path (about 4 s of dgemm on the host)."""
    N, m = 2000, 8
    mean0, diag0, lin, ang, idx, zr, zb = orc.synthetic_stream(N, 2, m, 1)
    cfg = orc.EkfConfig()
    om, oP = orc.ekf_step_dense(mean0, np.diag(diag0), lin[0], ang[0], idx[0], zr[0], zb[0], cfg)
    with sd.EkfSlam(len(mean0)) as f:
        f.set_state_diag(mean0, diag0)
        f.step(lin[0], ang[0], idx[0], zr[0], zb[0])
        mu, P = f.state()
    close(mu, om)
    close(P, oP)


def test_batched_trajectories_independent(sd):
    """Each trajectory of a batch equals its own single-trajectory oracle run (different seeds, and
    a different number of observations per trajectory per step)."""
    N, steps, B = 60, 15, 5
    cfg = orc.EkfConfig()
    streams = [orc.synthetic_stream(N, steps, 8, t) for t in range(B)]
    n = 3 + 2 * N
    with sd.EkfSlam(n, batch=B) as f:
        for b, s in enumerate(streams):
            f.set_state_diag(s[0], s[1], b)
        ref = [(s[0].copy(), np.diag(s[1])) for s in streams]
        for k in range(steps):
            mk = [(3 * b + k) % 9 for b in range(B)]           # 0..8 observations
            f.step([s[2][k] for s in streams], [s[3][k] for s in streams],
                   [s[4][k][:mk[b]] for b, s in enumerate(streams)],
                   [s[5][k][:mk[b]] for b, s in enumerate(streams)],
                   [s[6][k][:mk[b]] for b, s in enumerate(streams)])
            for b, s in enumerate(streams):
                ref[b] = orc.ekf_step_dense(ref[b][0], ref[b][1], s[2][k], s[3][k], s[4][k][:mk[b]],
                                            s[5][k][:mk[b]], s[6][k][:mk[b]], cfg)
        for b in range(B):
            mu, P = f.state(b)
            close(mu, ref[b][0])
            close(P, ref[b][1])


def test_predict_and_update_separately(sd):
    N = 40
    mean0, diag0, lin, ang, idx, zr, zb = orc.synthetic_stream(N, 20, 6, 2)
    cfg = orc.EkfConfig()
    om, oP = mean0.copy(), np.diag(diag0)
    with sd.EkfSlam(len(mean0)) as f:
        f.set_state_diag(mean0, diag0)
        for k in range(20):
            f.predict(lin[k], ang[k])
            om, oP = orc.predict_dense(om, oP, lin[k], ang[k], cfg)
            mu, P = f.state()
            close(mu, om)
            close(P, oP)
            f.update(idx[k], zr[k], zb[k])
            om, oP = orc.update_dense(om, oP, idx[k], zr[k], zb[k], cfg)
            mu, P = f.state()
            close(mu, om)
            close(P, oP)


def test_more_than_sixteen_landmarks_in_one_step(sd):
    """Lists longer than EKF_MMAX are split into device passes; the order is kept."""
    N, m = 64, 41
    mean0, diag0, lin, ang, idx, zr, zb = orc.synthetic_stream(N, 4, m, 5)
    cfg = orc.EkfConfig()
    om, oP = mean0.copy(), np.diag(diag0)
    with sd.EkfSlam(len(mean0)) as f:
        f.set_state_diag(mean0, diag0)
        for k in range(4):
            f.step(lin[k], ang[k], idx[k], zr[k], zb[k])
            om, oP = orc.ekf_step_dense(om, oP, lin[k], ang[k], idx[k], zr[k], zb[k], cfg)
        mu, P = f.state()
    close(mu, om)
    close(P, oP)


@pytest.mark.parametrize("flags", [dict(enable_measurement_model=False), dict(disable_motion_model=True),
                                   dict(enable_circular_interpolation=False),
                                   dict(motion_sigma=0.03, meas_sigma=0.2)])
def test_config_flags(sd, flags, both_paths):
    N = 30
    mean0, diag0, lin, ang, idx, zr, zb = orc.synthetic_stream(N, 25, 5, 7)
    ang = ang * 30.0            # theta crosses +-pi: wrap (:397) on, no wrap in the linear mode
    ocfg = orc.EkfConfig(**flags)
    om, oP = mean0.copy(), np.diag(diag0)
    with sd.EkfSlam(len(mean0), config=sd.EkfConfig(**flags)) as f:
        f.set_state_diag(mean0, diag0)
        for k in range(25):
            f.step(lin[k], ang[k], idx[k], zr[k], zb[k])
            om, oP = orc.ekf_step_dense(om, oP, lin[k], ang[k], idx[k], zr[k], zb[k], ocfg)
        mu, P = f.state()
        assert path_ran(f, both_paths)
    close(mu, om)
    close(P, oP)


def test_augmentation_matches_reference_growth(sd, both_paths):
    """add_landmarks == the zero-pad + 1e4 diagonal of :341-360, including cross terms staying zero."""
    rng = np.random.default_rng(3)
    n0 = 3 + 2 * 4
    A = rng.normal(size=(n0, n0))
    P0 = A @ A.T
    mu0 = rng.normal(size=n0)
    xy = rng.normal(size=(3, 2))
    with sd.EkfSlam(3 + 2 * 10) as f:
        f.set_state(mu0, P0)
        f.add_landmarks(xy)
        mu, P = f.state()
        f.step(0.0, 0.0, [], [], [])                    # (one launch, so that the path shows; the state was read before it)
        assert path_ran(f, both_paths)
    tp = {4 + i: [xy[i, 0], xy[i, 1]] for i in range(3)}
    om, oP = orc.augment(mu0, P0, 7, tp, orc.EkfConfig())
    assert np.array_equal(mu, om) and np.array_equal(P, oP)


def test_upload_download_roundtrip_bitexact(sd):
    rng = np.random.default_rng(0)
    n = 3 + 2 * 37
    A = rng.normal(size=(n, n))
    P0 = A + A.T
    mu0 = rng.normal(size=n)
    with sd.EkfSlam(3 + 2 * 50, batch=2) as f:
        f.set_state(mu0, P0, 1)
        mu, P = f.state(1)
        assert np.array_equal(mu, mu0) and np.array_equal(P, P0)
        assert f.size(0) == 3 and f.size(1) == n
        assert np.array_equal(f.covariance(0), np.eye(3) * 0.1)      # reference start, :69-70
        # a covariance is symmetric: the upper triangle of what the host hands over is authoritative
        f.set_state(mu0, A, 1)
        assert np.array_equal(f.covariance(1), np.triu(A) + np.triu(A, 1).T)
        assert np.array_equal(f.covariance_block(5, 2, 4, 3, 1), (np.triu(A) + np.triu(A, 1).T)[5:9, 2:5])


@pytest.mark.parametrize("n", [3, 43, 1003])
def test_predict_dense_mfma(sd, n):
    """P <- F P F^T + Q with a general dense F (fp64 MFMA GEMMs) vs NumPy dgemm."""
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n)) / np.sqrt(n)
    P0 = A @ A.T + np.eye(n)
    F = np.eye(n) + rng.normal(size=(n, n)) * 0.1          # asymmetric on purpose
    Nq = 0.01 * rng.normal(size=(n, n))
    Q = np.diag(rng.uniform(0.01, 0.1, n)) + Nq + Nq.T          # a covariance: symmetric
    with sd.EkfSlam(n) as f:
        f.set_state(np.zeros(n), P0)
        f.predict_dense(F, Q)
        P = f.covariance()
    close(P, F @ P0 @ F.T + Q, 1e-12)


def test_proto3_surface(sd, both_paths):
    """predict()/update() of src/EKF-SLAM.py:29-84 against vectors from the reference itself."""
    g = gu.load("proto3")
    state = np.array([0.0, 0.0, 0.0])
    cov = np.eye(3) * 0.1
    for k in range(len(g["dt"])):
        state, cov = sd.predict(state, cov, tuple(g["control"][k]), g["dt"][k])
        assert np.allclose(state, g["pred_state"][k], rtol=1e-12, atol=1e-14)
        close(cov, g["pred_cov"][k])
        state, cov = sd.update(state, cov, tuple(g["observation"][k]), g["landmark"][k])
        assert np.allclose(state, g["upd_state"][k], rtol=1e-9, atol=1e-12)
        close(cov, g["upd_cov"][k])
    from slam_duckietown_amd import ekf_bindings as eb
    assert path_ran(eb._proto["update"], both_paths)      # (`predict` is the dense product: no step kernel on either path)


def test_q_zero_propagates_nan_and_sets_flag(sd, both_paths):
    """Landmark exactly at the robot: q = 0 -> NaN like NumPy (:466-469), no trap, sticky flag."""
    mu = np.array([0.0, 0.0, 0.0, 0.0, 0.0])
    with sd.EkfSlam(5) as f:
        f.set_state(mu, np.eye(5))
        f.update([0], [0.5], [0.1])
        out = f.mean()
        assert not np.isfinite(out).all()
        assert f.flags() & 1 and path_ran(f, both_paths)


def test_argument_errors(sd, both_paths):
    with sd.EkfSlam(3 + 2 * 5) as f:
        f.set_state_diag(np.zeros(13), np.ones(13))
        with pytest.raises(sd.EkfError):
            f.update([1, 1], [0.5, 0.5], [0.0, 0.0])        # duplicate index
        with pytest.raises(sd.EkfError):
            f.update([5], [0.5], [0.0])                      # beyond the state
        with pytest.raises(sd.EkfError):
            f.set_state(np.zeros(15), np.eye(15))            # larger than n_max
    with pytest.raises(sd.EkfError):
        sd.EkfSlam(12)                                       # n_max must be 3 + 2N


@pytest.mark.parametrize("every", [1, 2, 3, 4, 9])
def test_deferred_covariance_pass(sd, every):
    """The covariance lives as P_base + pending low-rank factors; whatever the flush cadence, mean and
    covariance equal the dense reference path.  Steps mix 0..20 observations (0 = prediction only on
    top of pending ranks, > 16 = two device passes) so that rank slots fill unevenly."""
    N, steps = 70, 26
    mean0, diag0, lin, ang, idx, zr, zb = orc.synthetic_stream(N, steps, 20, 11)
    ms = [8, 3, 0, 20, 1, 8, 8, 0, 0, 5, 16, 2, 8, 8, 8, 8, 0, 4, 17, 8, 1, 1, 8, 6, 0, 8]
    cfg = orc.EkfConfig()
    om, oP = mean0.copy(), np.diag(diag0)
    with sd.EkfSlam(len(mean0)) as f:
        f.set_option("flush_every", every)
        f.set_state_diag(mean0, diag0)
        for k in range(steps):
            mk = ms[k]
            f.step(lin[k], ang[k], idx[k][:mk], zr[k][:mk], zb[k][:mk])
            om, oP = orc.ekf_step_dense(om, oP, lin[k], ang[k], idx[k][:mk], zr[k][:mk], zb[k][:mk], cfg)
            close(f.mean(), om)                      # the mean never waits for a flush
            if k in (6, 13, 25):
                close(f.covariance(), oP)            # reading the covariance flushes
        mu, P = f.state()
    close(mu, om)
    close(P, oP)


def test_deferred_pass_batched_n500(sd):
    """N=500, 3 trajectories, default cadence (4 steps per covariance pass), 10 steps (odd tail)."""
    N, steps, B = 500, 10, 3
    cfg = orc.EkfConfig()
    streams = [orc.synthetic_stream(N, steps, 8, 20 + t) for t in range(B)]
    ref = []
    for s in streams:
        om, oP = s[0].copy(), np.diag(s[1])
        for k in range(steps):
            om, oP = orc.ekf_step_structured(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], cfg)
        ref.append((om, oP))
    with sd.EkfSlam(3 + 2 * N, batch=B) as f:
        for b, s in enumerate(streams):
            f.set_state_diag(s[0], s[1], b)
        f.run_stream(np.stack([s[2] for s in streams], 1), np.stack([s[3] for s in streams], 1),
                     np.stack([s[4] for s in streams], 1), np.stack([s[5] for s in streams], 1),
                     np.stack([s[6] for s in streams], 1))
        for b in range(B):
            mu, P = f.state(b)
            close(mu, ref[b][0])
            close(P, ref[b][1])


def test_active_bound_is_exact(sd):
    """State indices beyond the highest landmark observed so far have exactly zero cross-covariance
    (block-diagonal start, BASELINE config 5): skipping them must change nothing, bit for bit."""
    N, steps = 300, 14
    mean0, diag0, lin, ang, idx, zr, zb = orc.synthetic_stream(N, steps, 8, 4)
    idx = (idx * 7 + 3) % 120          # observations wander inside the first 120 landmarks only
    for k in range(steps):             # (keep each step's indices distinct)
        assert len(set(idx[k])) == 8
    out = []
    for bound in (1, 0):
        with sd.EkfSlam(len(mean0)) as f:
            f.set_option("active_bound", bound)
            f.set_state_diag(mean0, diag0)
            f.run_stream(lin, ang, idx, zr, zb)
            out.append(f.state())
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    cfg = orc.EkfConfig()
    om, oP = mean0.copy(), np.diag(diag0)
    for k in range(steps):
        om, oP = orc.ekf_step_dense(om, oP, lin[k], ang[k], idx[k], zr[k], zb[k], cfg)
    close(out[0][0], om)
    close(out[0][1], oP)
    assert np.array_equal(out[0][1][243:, :243], np.zeros((len(mean0) - 243, 243)))   # never touched


def test_long_run_stays_on_the_reference(sd):
    """1600 steps at N=40 (every landmark revisited ~300 times) with the default deferred pass: no slow
    drift away from the dense reference path, no growth of the antisymmetric part of P.  Since round 6 these are 320
    CHAINED cadences (asserted): the block every solve starts from comes from the previous solve's records, and the pose
    block among them must be handed on as its upper triangle -- handed on as it was, its antisymmetric part grew x 1.16 per
    cadence and this test blew up past 1200 steps (profiles/r06_chained_solves.txt, 5 (e))."""
    N, steps, m = 40, 1600, 8
    mean0, diag0, lin, ang, idx, zr, zb = orc.synthetic_stream(N, steps, m, 6)
    cfg = orc.EkfConfig()
    om, oP = mean0.copy(), np.diag(diag0)
    for k in range(steps):
        om, oP = orc.ekf_step_dense(om, oP, lin[k], ang[k], idx[k], zr[k], zb[k], cfg)
    with sd.EkfSlam(len(mean0)) as f:
        f.set_state_diag(mean0, diag0)
        f.run_stream(lin, ang, idx, zr, zb)
        mu, P = f.state()
        assert f.flags() == 0
        assert sd.load_library().ekf_debug_chained(f._h) == steps * m // 40 - 1
    close(mu, om)
    close(P, oP)
    assert np.abs(P - P.T).max() <= 1e-11 * np.abs(P).max()


def test_max_size_n8000_three_steps(sd):
    """BASELINE config 5 size (n = 16003, P = 2.05 GB): three steps against the O(n^2) oracle, checked on
    the rows/columns the steps touch plus row/column sums (size-independent checksums of the rest)."""
    N, steps, m = 8000, 3, 8
    mean0, diag0, lin, ang, idx, zr, zb = orc.synthetic_stream(N, steps, m, 2)
    idx = (idx * 997 + 13) % N                       # scatter the observations over the whole state
    cfg = orc.EkfConfig()
    om, oP = mean0.copy(), np.diag(diag0)
    for k in range(steps):
        om, oP = orc.ekf_step_structured(om, oP, lin[k], ang[k], idx[k], zr[k], zb[k], cfg)
    n = len(mean0)
    with sd.EkfSlam(n) as f:
        f.set_option("active_bound", 0)              # exercise the full-size pass
        f.set_state_diag(mean0, diag0)
        f.run_stream(lin, ang, idx, zr, zb)
        mu = f.mean()
        rows = sorted({0, 1, 2, n - 1} | {3 + 2 * int(j) for j in idx.ravel()})
        for r in rows[:12]:
            close(f.covariance_block(r, 0, 1, n)[0], oP[r])
            close(f.covariance_block(0, r, n, 1)[:, 0], oP[:, r])
        P = f.covariance()
    close(mu, om)
    close(P.sum(axis=1), oP.sum(axis=1))
    close(P.sum(axis=0), oP.sum(axis=0))
    close(np.diag(P), np.diag(oP))
    del P, oP


@pytest.mark.parametrize("case", gu.REPLAY_CASES)
def test_device_side_association_golden(sd, case, both_paths):
    """association, gate, averaging and augmentation on the GPU (`step_detections`)
    against the reference's own outputs: state size, update order, TAG_INDEX, mean and covariance."""
    g = gu.load(case)
    cfg = sd.EkfConfig(enable_measurement_model=bool(g["flag_measurement"]),
                       enable_circular_interpolation=bool(g["flag_circular"]),
                       disable_motion_model=bool(g["flag_no_motion"]))
    with sd.EkfSlam(3 + 2 * 12, config=cfg) as f:
        if gu.ignore_tags(g):                      # IGNORE_TAGS (:36-37, :286) in the device's ignore table
            f.set_association(1.5, gu.ignore_tags(g))
        for k in range(len(g["lin"])):
            det = gu.detections_for_step(g, k)
            f.step_detections(g["lin"][k], g["ang"][k], det)
            n = int(g["out_size"][k])
            assert f.size() == n
            if k % 4 == 0 or k == len(g["lin"]) - 1:
                tp = f.tags_positions()
                assert list(tp.keys()) == [i for i in g["out_obs_order"][k] if i >= 0]
                mu, P = f.state()
                close(mu, g["out_mean"][k, :n])
                close(P, g["out_cov"][k, :n, :n])
        assert sorted(f.tag_index().items(), key=lambda kv: kv[1]) == [tuple(r) for r in g["out_tag_index"]]
        assert f.flags() == 0 and path_ran(f, both_paths)


def test_device_association_matches_host_association(sd, both_paths):
    """tags_positions from the device equal the host front end's (ulp-level differences only)."""
    g = gu.load("replay_default")
    ti = {}
    with sd.EkfSlam(3 + 2 * 12) as f:
        for k in range(20):
            det = gu.detections_for_step(g, k)
            pose = f.mean()[:3]
            host = sd.associate(det, ti, pose)
            f.step_detections(g["lin"][k], g["ang"][k], det)
            dev = f.tags_positions()
            assert list(dev.keys()) == list(host.keys())
            for key in host:
                assert dev[key][3] == host[key][3]
                assert np.allclose([dev[key][i] for i in (0, 1, 2, 4, 5)], [host[key][i] for i in (0, 1, 2, 4, 5)],
                                   rtol=1e-13, atol=1e-15)
        assert f.tag_index() == ti and path_ran(f, both_paths)


def test_device_association_overflow_flag(sd, both_paths):
    from types import SimpleNamespace as NS
    mk = lambda i, x, z: NS(tag_id=i, pose_R=np.eye(3), pose_t=np.array([[x], [0.0], [z]]), pose_err=0.0)
    with sd.EkfSlam(3 + 2 * 2) as f:                    # room for two landmarks only
        f.step_detections(0.01, 0.0, [(0.0, [mk(5, 0.1, 0.5), mk(6, -0.1, 0.6), mk(7, 0.0, 0.7)])])
        assert f.size() == 7 and f.tag_index() == {5: 0, 6: 1}
        assert f.flags() & 2                              # EKF_FLAG_ASSOC: the third tag did not fit
        assert path_ran(f, both_paths)


def test_scattered_landmarks_varying_m(sd):
    """Observed landmarks scattered over the whole map, 1..16 per step, two trajectories with different lists:
    every wave of the panel kernel sees gathered rows below, inside and above its own 64 state indices, and
    steps of different rank counts are packed into one pending update."""
    N, steps, B = 700, 14, 2
    n = 3 + 2 * N
    cfg = orc.EkfConfig()
    rng = np.random.default_rng(11)
    world = [orc.synthetic_world(N, t) for t in range(B)]
    om = [w[2].copy() for w in world]
    oP = [np.diag(w[3]) for w in world]
    pose = [np.zeros(3) for _ in range(B)]
    with sd.EkfSlam(n, batch=B) as f:
        for b in range(B):
            f.set_state_diag(world[b][2], world[b][3], b)
        for k in range(steps):
            m_b = [int(rng.integers(1, 17)), int(rng.integers(1, 17))]
            lin, ang = 0.004 + 0.001 * k, (0.02 if k % 3 else 0.004)
            idx, zr, zb = [], [], []
            for b in range(B):
                pose[b], _ = orc.motion_model(pose[b], lin, ang, cfg)
                vis = rng.choice(N, size=m_b[b], replace=False)
                d = world[b][1][vis] - pose[b][0:2]
                cth, sth = np.cos(pose[b][2]), np.sin(pose[b][2])
                xr = cth * d[:, 0] + sth * d[:, 1] + rng.normal(0, 0.01, m_b[b])
                yr = -sth * d[:, 0] + cth * d[:, 1] + rng.normal(0, 0.01, m_b[b])
                idx.append(vis)
                zr.append(np.sqrt(xr ** 2 + yr ** 2))
                zb.append(np.arctan2(yr, xr))
                om[b], oP[b] = orc.ekf_step_structured(om[b], oP[b], lin, ang, vis, zr[b], zb[b], cfg)
            f.step([lin] * B, [ang] * B, idx, zr, zb)
            if k in (0, 5, steps - 1):
                for b in range(B):
                    mu, P = f.state(b)
                    close(mu, om[b])
                    close(P, oP[b])
                    assert np.array_equal(P, P.T)
        assert f.flags(0) == 0 and f.flags(1) == 0


@pytest.mark.parametrize("N,B,m", [(300, 2, 8), (531, 3, 5), (64, 9, 8), (1100, 2, 3)])
def test_row_slab_pass_is_bit_identical(sd, N, B, m):
    """`pass_kernel=2` (k_flush_rs: W fragments in registers, the V strip shared through LDS, software-pipelined
    tiles, persistent workgroups on a work queue) applies the same update as k_flush, bit for bit: all rank counts
    (4..20 k-tiles), both cache policies, slabs that end inside a 128-row block, states smaller than one slab."""
    steps = 11
    n = 3 + 2 * N
    streams = [orc.synthetic_stream(N, steps, m, 30 + t) for t in range(B)]
    starts = []
    for t in range(B):
        rng = np.random.default_rng(90 + t)
        A = rng.normal(size=(n, 6)) * 0.3
        starts.append(A @ A.T + np.diag(rng.uniform(0.5, 2.0, n)))
    out = {}
    for kernel in (0, 2):
        for limit, streaming in ((2 * m, 0), (6 * m, 1), (80, 1), (80, 0)):
            with sd.EkfSlam(n, batch=B) as f:
                f.set_option("pass_kernel", kernel)
                f.set_option("rank_limit", limit)
                f.set_option("pass_streaming", streaming)
                f.set_option("active_bound", 0)
                for b, s in enumerate(streams):
                    f.set_state(s[0], starts[b], b)
                for k in range(steps):
                    f.step([s[2][k] for s in streams], [s[3][k] for s in streams], [s[4][k] for s in streams],
                           [s[5][k] for s in streams], [s[6][k] for s in streams])
                out[kernel, limit, streaming] = [f.state(b) for b in range(B)]
                assert [f.flags(b) for b in range(B)] == [0] * B
    for key in [k[1:] for k in out if k[0] == 0]:
        for b in range(B):
            assert np.array_equal(out[(0,) + key][b][0], out[(2,) + key][b][0])
            assert np.array_equal(out[(0,) + key][b][1], out[(2,) + key][b][1]), (key, b)
    # and the result itself is the reference's
    cfg = orc.EkfConfig()
    s = streams[0]
    om, oP = s[0].copy(), starts[0].copy()
    for k in range(steps):
        om, oP = orc.ekf_step_dense(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], cfg)
    close(out[2, 80, 1][0][0], om)
    close(out[2, 80, 1][0][1], oP)


@pytest.mark.parametrize("B", [8, 10, 13, 16, 24])
def test_row_slab_queue_modes_on_the_device(sd, B):
    """The hand-out orders of the row-slab pass that the automatic rule picks by batch size (launch_flush_rs_t), checked
    for VALUES on the device, not only as host enumerations: 8 trajectories (mode 1, every queue's only trajectory cut
    into chunks), 10 (mode 3: dealt half slabs), 13 (mode 2: dealt whole slabs), 16 (mode 1, pairs), 24 (mode 1: a pair
    plus a chunked lone trajectory per queue) -- each bit for bit the column-strip kernel's result."""
    N, m, steps = 300, 8, 6                        # n = 603: 5 slabs of up to 10 strips (mode 3 needs >= 8 strips)
    n = 3 + 2 * N
    streams = [orc.synthetic_stream(N, steps, m, 130 + t) for t in range(B)]
    starts = []
    for t in range(B):
        rng = np.random.default_rng(290 + t)
        A = rng.normal(size=(n, 4)) * 0.3
        starts.append(A @ A.T + np.diag(rng.uniform(0.5, 2.0, n)))
    out = {}
    for kernel in (0, 2):
        with sd.EkfSlam(n, batch=B) as f:
            f.set_option("pass_kernel", kernel)        # (2 with pass_chunk = 0: the automatic hand-out order)
            f.set_option("pass_streaming", 1)
            f.set_option("active_bound", 0)
            for b, s in enumerate(streams):
                f.set_state(s[0], starts[b], b)
            for k in range(steps):
                f.step([s[2][k] for s in streams], [s[3][k] for s in streams], [s[4][k] for s in streams],
                       [s[5][k] for s in streams], [s[6][k] for s in streams])
            f.flush()
            assert f.last_pass().startswith("ekf::k_flush_rs" if kernel == 2 else "ekf::k_flush<")
            out[kernel] = [f.state(b) for b in range(B)]
            assert [f.flags(b) for b in range(B)] == [0] * B
    for b in range(B):
        assert np.array_equal(out[0][b][0], out[2][b][0]), b
        assert np.array_equal(out[0][b][1], out[2][b][1]), b
    cfg = orc.EkfConfig()
    s = streams[B - 1]
    om, oP = s[0].copy(), starts[B - 1].copy()
    for k in range(steps):
        om, oP = orc.ekf_step_dense(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], cfg)
    close(out[2][B - 1][0], om)
    close(out[2][B - 1][1], oP)


def test_row_slab_pass_with_active_bound_and_growing_state(sd):
    """k_flush_rs on a block-diagonal start with the active bound on (slabs beyond the bound are skipped, the last slab
    ends inside a block) and with trajectories of different sizes in one batch."""
    N, steps, m, B = 400, 14, 8, 3
    streams = [orc.synthetic_stream(N, steps, m, 50 + t) for t in range(B)]
    sizes = [3 + 2 * N, 3 + 2 * 150, 3 + 2 * 333]
    out = {}
    for kernel in (0, 2):
        with sd.EkfSlam(3 + 2 * N, batch=B) as f:
            f.set_option("pass_kernel", kernel)
            f.set_option("pass_streaming", 1)
            for b, s in enumerate(streams):
                f.set_state_diag(s[0][:sizes[b]], s[1][:sizes[b]], b)
            for k in range(steps):
                idx = [(s[4][k] * 3 + 1) % ((sizes[b] - 3) // 2 // 2) for b, s in enumerate(streams)]
                f.step([s[2][k] for s in streams], [s[3][k] for s in streams], idx,
                       [s[5][k] for s in streams], [s[6][k] for s in streams])
            out[kernel] = [f.state(b) for b in range(B)]
    for b in range(B):
        assert np.array_equal(out[0][b][0], out[2][b][0]) and np.array_equal(out[0][b][1], out[2][b][1])


def test_pass_scheduling_knobs_do_not_change_the_result(sd):
    """How the covariance pass is cut into work -- strips per unit of the row-slab kernel (`pass_chunk`), its number of
    persistent workgroups (`pass_workgroups`: fewer than units, so that the queues and the stealing are exercised),
    rows per workgroup of the column-strip kernel (`pass_rows_per_block`, also what the automatic rule varies) --
    never changes a bit of mean or covariance."""
    N, B, m, steps = 531, 3, 5, 9
    n = 3 + 2 * N
    streams = [orc.synthetic_stream(N, steps, m, 60 + t) for t in range(B)]
    starts = []
    for t in range(B):
        rng = np.random.default_rng(190 + t)
        A = rng.normal(size=(n, 5)) * 0.3
        starts.append(A @ A.T + np.diag(rng.uniform(0.5, 2.0, n)))

    def run(options):
        with sd.EkfSlam(n, batch=B) as f:
            f.set_option("active_bound", 0)
            for name, value in options:
                f.set_option(name, value)
            for b, s in enumerate(streams):
                f.set_state(s[0], starts[b], b)
            for k in range(steps):
                f.step([s[2][k] for s in streams], [s[3][k] for s in streams], [s[4][k] for s in streams],
                       [s[5][k] for s in streams], [s[6][k] for s in streams])
            assert [f.flags(b) for b in range(B)] == [0] * B
            return [f.state(b) for b in range(B)]

    ref = run([("pass_kernel", 0)])
    variants = [[("pass_kernel", 2), ("pass_chunk", c)] for c in (1, 2, 5, 64)]
    variants += [[("pass_kernel", 2), ("pass_workgroups", w)] for w in (1, 7, 100)]
    variants += [[("pass_kernel", 2), ("pass_chunk", 3), ("pass_workgroups", 5), ("pass_streaming", 1)]]
    variants += [[("pass_kernel", 0), ("pass_rows_per_block", r)] for r in (16, 64, 160, 256)]
    for opts in variants:
        got = run(opts)
        for b in range(B):
            assert np.array_equal(got[b][0], ref[b][0]), opts
            assert np.array_equal(got[b][1], ref[b][1]), opts


def test_row_slab_pass_on_equal_static_shares_is_bit_identical(sd):
    """The row-slab pass for a few long trajectories: one equal static share of the strips per workgroup, pieces that
    start and end inside slabs (launch_flush_rs with a share table; automatic at N = 8000 x 1, forced here at a size the
    oracle handles through `pass_workgroups`).  Trajectories of different sizes, dense and block-diagonal starts, 80 and
    16 ranks -- bit for bit the column-strip kernel's result, and the reference's."""
    import ctypes as C
    lib = sd.load_library()
    N, B, m, steps = 700, 3, 8, 6
    n = 3 + 2 * N
    sizes = [n, 3 + 2 * 450, n]
    streams = [orc.synthetic_stream(N, steps, m, 160 + t) for t in range(B)]
    starts = []
    for t in range(B):
        rng = np.random.default_rng(390 + t)
        A = rng.normal(size=(sizes[t], 4)) * 0.3
        starts.append(A @ A.T + np.diag(rng.uniform(0.5, 2.0, sizes[t])))
    idx = [np.stack([s[4][k] % ((sizes[b] - 3) // 2) for k in range(steps)]) for b, s in enumerate(streams)]

    def run(options, dense):
        with sd.EkfSlam(n, batch=B) as f:
            f.set_option("pass_streaming", 1)
            f.set_option("active_bound", 0 if dense else 1)
            for name, value in options:
                f.set_option(name, value)
            for b, s in enumerate(streams):
                if dense:
                    f.set_state(s[0][:sizes[b]], starts[b], b)
                else:
                    f.set_state_diag(s[0][:sizes[b]], s[1][:sizes[b]], b)
            used = []
            for k in range(steps):
                f.step([s[2][k] for s in streams], [s[3][k] for s in streams], [idx[b][k] for b in range(B)],
                       [s[5][k] for s in streams], [s[6][k] for s in streams])
                if k in (4, 5):
                    f.flush()
                    used.append(lib.ekf_debug_last_pass_shares(C.c_void_p(f._h.value)))
            assert [f.flags(b) for b in range(B)] == [0] * B
            return [f.state(b) for b in range(B)], used

    for dense in (True, False):
        ref, used0 = run([("pass_kernel", 0)], dense)
        assert used0 == [0, 0]
        for wgs in (8, 5):
            got, used = run([("pass_kernel", 2), ("pass_workgroups", wgs)], dense)
            if dense:                                          # (block-diagonal start: the active bound is still small,
                assert all(u >= 1 for u in used), used         #  the queue modes apply) both passes ran on share tables
            for b in range(B):
                assert np.array_equal(got[b][0], ref[b][0]) and np.array_equal(got[b][1], ref[b][1]), (dense, wgs, b)
    cfg = orc.EkfConfig()
    s = streams[1]
    om, oP = s[0][:sizes[1]].copy(), starts[1].copy()
    for k in range(steps):
        om, oP = orc.ekf_step_dense(om, oP, s[2][k], s[3][k], idx[1][k], s[5][k], s[6][k], cfg)
    got, _ = run([("pass_kernel", 2), ("pass_workgroups", 8)], True)
    close(got[1][0], om)
    close(got[1][1], oP)


def test_static_shares_for_ten_to_fourteen_trajectories(sd):
    """10 .. 14 trajectories take the row-slab pass on equal static shares as well (the queue modes cannot balance one
    to two slabs per workgroup): bit for bit the column-strip kernel's result.  `pass_workgroups` = 6 makes the shares
    long enough at a size the test can download (automatic from N ~ 1700 on the full chip)."""
    import ctypes as C
    lib = sd.load_library()
    N, B, m, steps = 300, 12, 8, 11
    n = 3 + 2 * N
    streams = [orc.synthetic_stream(N, steps, m, 260 + t) for t in range(B)]
    args = [np.stack([s[i] for s in streams], axis=1) for i in (2, 3, 4, 5, 6)]
    res = {}
    for kernel, wgs in ((0, 0), (2, 6)):
        with sd.EkfSlam(n, batch=B) as f:
            f.set_option("pass_streaming", 1)
            f.set_option("active_bound", 0)
            f.set_option("lookahead", 0)         # (the two FORMS OF THE PASS are compared: the same order of launches around both --
            f.set_option("pass_kernel", kernel)  #  the column-strip pass of a bank this small would have its solves chained)
            if wgs:
                f.set_option("pass_workgroups", wgs)
            for b, s in enumerate(streams):
                f.set_state_diag(s[0], s[1], b)
            f.run_stream(*args)
            f.flush()
            res[kernel] = [f.state(b) for b in range(B)]
            assert [f.flags(b) for b in range(B)] == [0] * B
            shares = lib.ekf_debug_last_pass_shares(C.c_void_p(f._h.value))
            assert (shares >= 1) if kernel == 2 else (shares == 0), (kernel, shares)
    for b in range(B):
        assert np.array_equal(res[2][b][0], res[0][b][0]) and np.array_equal(res[2][b][1], res[0][b][1]), b


def test_single_launch_step_is_bit_identical(sd):
    """`fused_step=1` (k_step_split: solve and panel workgroups in one launch, the panels gathered beside the solve and
    released by a per-trajectory step counter) gives the results of the two-launch path bit for bit: ragged observation
    counts (0..16, so every instantiation runs), scattered landmarks, three trajectories of different sizes with the
    active bound on, flags clean (EKF_FLAG_INTERNAL would mean a wait timed out)."""
    N, steps, B = 300, 17, 3
    rng = np.random.default_rng(5)
    world = [orc.synthetic_world(N, 40 + t) for t in range(B)]
    sizes = [3 + 2 * N, 3 + 2 * 120, 3 + 2 * 211]
    ms = [[int(rng.integers(0, 17)) for _ in range(B)] for _ in range(steps)]
    obs = []
    for k in range(steps):
        row = []
        for b in range(B):
            nl = (sizes[b] - 3) // 2
            vis = rng.choice(nl, size=ms[k][b], replace=False)
            row.append((vis, rng.uniform(0.3, 1.4, ms[k][b]), rng.uniform(-1.0, 1.0, ms[k][b])))
        obs.append(row)
    out = {}
    for fused in (0, 1):
        with sd.EkfSlam(3 + 2 * N, batch=B) as f:
            f.set_option("fused_step", fused)
            for b in range(B):
                f.set_state_diag(world[b][2][:sizes[b]], world[b][3][:sizes[b]], b)
            for k in range(steps):
                f.step([0.004 + 0.001 * k] * B, [0.02 if k % 3 else 0.004] * B, [o[0] for o in obs[k]],
                       [o[1] for o in obs[k]], [o[2] for o in obs[k]])
                if k == 7:
                    out[fused, "mid"] = [f.mean(b) for b in range(B)]
            out[fused, "end"] = [f.state(b) for b in range(B)]
            assert [f.flags(b) for b in range(B)] == [0] * B
    for b in range(B):
        assert np.array_equal(out[0, "mid"][b], out[1, "mid"][b])
        assert np.array_equal(out[0, "end"][b][0], out[1, "end"][b][0])
        assert np.array_equal(out[0, "end"][b][1], out[1, "end"][b][1])


def test_single_launch_step_throughput_shape_is_bit_identical(sd):
    """Mid-size batches (more than 512 waves of state indices, but room on the chip for one more workgroup per
    trajectory) run a step as k_panels_split: workgroup 0 of a trajectory solves, the others gather their panels
    meanwhile and fetch the solve's header and records from the mailbox.  Same results as the two-launch path bit for
    bit (16 trajectories of different sizes, active bound on and off), and the reference's for one of them."""
    N, B, m, steps = 1200, 16, 8, 8
    streams = [orc.synthetic_stream(N, steps, m, 70 + t) for t in range(B)]
    sizes = [3 + 2 * (N - 37 * (t % 5)) for t in range(B)]
    out = {}
    for bound in (0, 1):
        for fused in (0, 1):
            with sd.EkfSlam(3 + 2 * N, batch=B) as f:
                f.set_option("fused_step", fused)
                f.set_option("active_bound", bound)
                for b, s in enumerate(streams):
                    f.set_state_diag(s[0][:sizes[b]], s[1][:sizes[b]], b)
                for k in range(steps):
                    idx = [s[4][k] % ((sizes[b] - 3) // 2) for b, s in enumerate(streams)]
                    f.step([s[2][k] for s in streams], [s[3][k] for s in streams], idx,
                           [s[5][k] for s in streams], [s[6][k] for s in streams])
                out[bound, fused] = [f.state(b) for b in (0, 3, 7, 15)] + [f.mean(b) for b in range(B)]
                assert [f.flags(b) for b in range(B)] == [0] * B
        for x, y in zip(out[bound, 0], out[bound, 1]):
            if isinstance(x, tuple):
                assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
            else:
                assert np.array_equal(x, y)
    cfg = orc.EkfConfig()
    s = streams[0]
    om, oP = s[0].copy(), np.diag(s[1])
    for k in range(steps):
        om, oP = orc.ekf_step_structured(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], cfg)
    close(out[0, 1][0][0], om)
    close(out[0, 1][0][1], oP)
