"""GPU parity of the SMALL-STATE path (slam-duckietown_amd/csrc/ekf_small.hip): filters whose covariance fits a CU's LDS
(n_max <= 79: up to 38 landmarks; the reference's real map has 12, src/replay_no_ros.py:26) run every step -- or a whole
uploaded stream -- inside one workgroup with P resident in LDS.  Same arithmetic as the reference (simple-form update,
sequential re-linearisation, src/replay_no_ros.py:368-480), so the cases are the reference's own golden vectors: the N = 20
streams (BASELINE config 1) step by step and as one launch, q = 0, augmentation, banks of trajectories of different size,
the step-and-fetch entry point -- each asserted to have taken the small-state kernel.  (The whole-function replay fixtures
through the drop-in and through the device-side association, the flag variants, the 3-state surface, the replay driver, the
node adapter and the Monte-Carlo evaluation run on BOTH paths where they live: the `both_paths` fixture of tests/conftest.py.)
"""
import ctypes as C

import numpy as np
import pytest

from oracle import ekf_oracle as orc
from tests import golden_util as gu

pytestmark = pytest.mark.gpu

REL_TOL = 1e-6
TIGHT = 1e-9


@pytest.fixture(scope="module")
def sd():
    import slam_duckietown_amd as sd
    sd.load_library()
    return sd


def close(a, b, tol=TIGHT):
    r = orc.rel_fro(a, b)
    assert r < REL_TOL, f"rel Frobenius {r:.3e} exceeds the 1e-6 bar"
    assert r < tol, f"rel Frobenius {r:.3e} exceeds the expected {tol:g}"


def small_launches(sd, f):
    lib = sd.load_library()
    return lib.ekf_debug_small_launches(f._h)


def test_small_state_is_the_default_up_to_64_landmarks(sd, monkeypatch):
    monkeypatch.delenv("EKFSLAM_HIP_SMALL_STATE", raising=False)
    s = orc.synthetic_stream(20, 3, 8, 0)
    for n_max, expect in ((43, 3), (79, 3), (81, 0)):
        with sd.EkfSlam(n_max) as f:
            f.set_state_diag(s[0], s[1])
            for k in range(3):
                f.step(s[2][k], s[3][k], s[4][k], s[5][k], s[6][k])
            assert small_launches(sd, f) == expect
            f.set_option("small_state", 0)
            f.step(s[2][0], s[3][0], s[4][0], s[5][0], s[6][0])
            assert small_launches(sd, f) == expect


@pytest.mark.parametrize("case", ["stream_n20_m8", "stream_n20_m1"])
def test_golden_streams_step_by_step(sd, case):
    """The reference's own outputs, every step (BASELINE config 1: N = 20, 500 steps), one launch per step."""
    g = gu.load(case)
    n = len(g["mean0"])
    kept = {int(s): i for i, s in enumerate(g["out_cov_steps"])}
    with sd.EkfSlam(n) as f:
        f.set_option("small_state", 1)
        f.set_state_diag(g["mean0"], g["diag0"])
        for k in range(len(g["lin"])):
            f.step(g["lin"][k], g["ang"][k], g["idx"][k], g["zr"][k], g["zb"][k])
            close(f.mean(), g["out_mean"][k])
            if k in kept:
                P = f.covariance()
                close(P, g["out_cov"][kept[k]])
                assert np.array_equal(P, P.T)
        assert f.flags() == 0 and small_launches(sd, f) == len(g["lin"])


@pytest.mark.parametrize("case", ["stream_n20_m8", "stream_n20_m1"])
def test_golden_streams_as_one_launch(sd, case):
    """The same streams uploaded and run as ONE launch per piece (P stays in LDS across all steps of a piece), in pieces that
    end on the steps the fixture kept a covariance for; against the golden vectors and against the general kernels."""
    g = gu.load(case)
    n = len(g["mean0"])
    steps = len(g["lin"])
    cuts = sorted(set(int(s) + 1 for s in g["out_cov_steps"]) | {steps})
    kept = {int(s): i for i, s in enumerate(g["out_cov_steps"])}
    out = {}
    for small in (1, 0):
        with sd.EkfSlam(n) as f:
            f.set_option("small_state", small)
            f.set_state_diag(g["mean0"], g["diag0"])
            f.stream_upload(g["lin"], g["ang"], g["idx"], g["zr"], g["zb"])
            at = 0
            for c in cuts:
                f.stream_run(at, c - at)
                at = c
                mu, P = f.state()
                close(mu, g["out_mean"][c - 1])
                if c - 1 in kept:
                    close(P, g["out_cov"][kept[c - 1]])
            assert f.flags() == 0
            assert small_launches(sd, f) == (len(cuts) if small else 0)
            out[small] = (mu, P)
    close(out[1][0], out[0][0], 1e-10)
    close(out[1][1], out[0][1], 1e-10)


def test_bank_of_small_filters_of_different_size_against_the_oracle(sd):
    """12 trajectories of 5 .. 38 landmarks in one handle (one workgroup each), dense starts, predictions and updates on their
    own, 0 .. 20 observations per step (more than 16 are split into passes), an uploaded stream in the middle."""
    rng = np.random.default_rng(7)
    sizes = [5, 12, 12, 20, 22, 23, 24, 31, 33, 37, 38, 38]
    B = len(sizes)
    cfg = orc.EkfConfig()
    means, covs = [], []
    for b, N in enumerate(sizes):
        n = 3 + 2 * N
        A = rng.normal(size=(n, 5)) * 0.3
        covs.append(A @ A.T + np.diag(rng.uniform(0.5, 2.0, n)))
        means.append(np.concatenate([[0.0, 0.0, 0.1 * b], rng.uniform(-1.0, 1.0, 2 * N)]))

    def observe(b, m):
        N = sizes[b]
        idx = rng.choice(N, size=min(m, N), replace=False).astype(np.int32)
        dx, dy = means[b][3 + 2 * idx] - means[b][0], means[b][4 + 2 * idx] - means[b][1]
        return idx, np.hypot(dx, dy) + rng.normal(0, 0.02, len(idx)), np.arctan2(dy, dx) - means[b][2] + rng.normal(0, 0.02, len(idx))

    with sd.EkfSlam(79, batch=B) as f:
        f.set_option("small_state", 1)
        for b in range(B):
            f.set_state(means[b], covs[b], b)
        for it in range(12):
            lin, ang = rng.uniform(0.002, 0.02, B), rng.uniform(-0.3, 0.3, B)
            ang[rng.random(B) < 0.3] = 0.004
            obs = [observe(b, int(rng.integers(0, 21))) for b in range(B)]
            kind = it % 4
            if kind == 0:
                f.step(lin, ang, [o[0] for o in obs], [o[1] for o in obs], [o[2] for o in obs])
                for b in range(B):
                    means[b], covs[b] = orc.ekf_step_dense(means[b], covs[b], lin[b], ang[b], *obs[b], cfg)
            elif kind == 1:
                f.predict(lin, ang)
                for b in range(B):
                    means[b], covs[b] = orc.predict_dense(means[b], covs[b], lin[b], ang[b], cfg)
            elif kind == 2:
                f.update([o[0] for o in obs], [o[1] for o in obs], [o[2] for o in obs])
                for b in range(B):
                    means[b], covs[b] = orc.update_dense(means[b], covs[b], *obs[b], cfg)
            else:
                steps, mcap = 7, 4
                idx = np.zeros((steps, B, mcap), dtype=np.int32)
                zr, zb = np.zeros((steps, B, mcap)), np.zeros((steps, B, mcap))
                ms = np.zeros((steps, B), dtype=np.int32)
                lins, angs = rng.uniform(0.002, 0.02, (steps, B)), rng.uniform(-0.2, 0.2, (steps, B))
                for k in range(steps):
                    for b in range(B):
                        o = observe(b, int(rng.integers(0, mcap + 1)))
                        m = len(o[0])
                        ms[k, b] = m
                        idx[k, b, :m], zr[k, b, :m], zb[k, b, :m] = o
                        means[b], covs[b] = orc.ekf_step_dense(means[b], covs[b], lins[k, b], angs[k, b], o[0], o[1], o[2], cfg)
                f.run_stream(lins, angs, idx, zr, zb, ms)
        assert small_launches(sd, f) > 0
        for b in range(B):
            mu, P = f.state(b)
            assert f.flags(b) == 0 and len(mu) == 3 + 2 * sizes[b]
            close(mu, means[b])
            close(P, covs[b])
            assert np.array_equal(P, P.T)


def test_q_zero_and_growth_on_the_small_path(sd):
    """q = 0 (a landmark exactly at the robot's position) propagates NaN like NumPy's 0/0 at :466-469 and raises the sticky
    flag -- in its own trajectory only; augmentation between small-state steps."""
    with sd.EkfSlam(3 + 2 * 10, batch=2) as f:
        f.set_option("small_state", 1)
        mu = np.array([0.2, -0.1, 0.3, 0.2, -0.1, 1.0, 0.5])
        for b in range(2):
            f.set_state(mu, np.eye(7) * 0.2, b)
        f.add_landmarks(np.array([[0.4, 0.9]]), 1)
        f.step([0.0, 0.004], [0.0, 0.02], [[0], [2]], [[0.3], [0.9]], [[0.1], [0.2]])
        assert f.flags(0) & 1 and not np.isfinite(f.mean(0)).all()
        assert f.flags(1) == 0
        om, oP = np.concatenate([mu, [0.4, 0.9]]), np.zeros((9, 9))
        oP[:7, :7] = np.eye(7) * 0.2
        oP[7, 7] = oP[8, 8] = 1.0e4
        om, oP = orc.ekf_step_dense(om, oP, 0.004, 0.02, [2], [0.9], [0.2], orc.EkfConfig())
        m1, P1 = f.state(1)
        close(m1, om)
        close(P1, oP)


def fused_fetches(sd, f):
    lib = sd.load_library()
    return lib.ekf_debug_fused_fetches(f._h)


@pytest.mark.parametrize("small,batch,which", [(1, 1, 0), (1, 3, 1), (0, 1, 0), (0, 2, 1)])
def test_step_state_is_step_then_state(sd, small, batch, which):
    """`ekf_step_fetch` (one iteration of the reference's loop: step, then mean and covariance back,
    src/replay_no_ros.py:229-237): bit-identical to `step` + `state` on a twin filter, equal to the oracle, and on the
    small-state path answered by the step's own launch -- also when a long observation list is split into two launches, when
    nothing is observed, and for a trajectory that is not the first of its bank."""
    rng = np.random.default_rng(77 + small + batch)
    cfg = orc.EkfConfig()
    N = 24
    streams = [orc.synthetic_stream(N, 14, 8, 40 + t) for t in range(batch)]
    means = [s[0].copy() for s in streams]
    covs = [np.diag(s[1]) for s in streams]
    with sd.EkfSlam(3 + 2 * N, batch=batch) as f, sd.EkfSlam(3 + 2 * N, batch=batch) as twin:
        for g in (f, twin):
            g.set_option("small_state", small)
            for t, s in enumerate(streams):
                g.set_state_diag(s[0], s[1], t)
        for k in range(14):
            m = [0, 1, 8, 20, 3][k % 5]                       # 20: more than one launch (16 landmarks per launch)
            obs = []
            for t in range(batch):
                idx = rng.choice(N, size=m, replace=False).astype(np.int32)
                lx, ly = means[t][3 + 2 * idx], means[t][4 + 2 * idx]
                zr = np.hypot(lx - means[t][0], ly - means[t][1]) + rng.normal(0, 0.02, m)
                zb = np.arctan2(ly - means[t][1], lx - means[t][0]) - means[t][2] + rng.normal(0, 0.02, m)
                obs.append((idx, zr, zb))
            lin = np.array([s[2][k] for s in streams])
            ang = np.array([s[3][k] for s in streams])
            args = (lin, ang, [o[0] for o in obs], [o[1] for o in obs], [o[2] for o in obs])
            mu, P = f.step_state(*args, b=which)
            twin.step(*args)
            mu2, P2 = twin.state(which)
            for t in range(batch):
                means[t], covs[t] = orc.ekf_step_dense(means[t], covs[t], lin[t], ang[t], *obs[t], cfg)
            assert np.array_equal(mu, mu2) and np.array_equal(P, P2), k
            assert np.array_equal(P, P.T)
            close(mu, means[which])
            close(P, covs[which])
        assert f.flags(which) == 0
        assert fused_fetches(sd, f) == (14 if small else 0)
        assert fused_fetches(sd, twin) == 0
        # the state stayed resident: the other trajectories, and the same one read the ordinary way
        for t in range(batch):
            mu, P = f.state(t)
            close(mu, means[t])
            close(P, covs[t])


def test_step_state_reports_what_the_two_calls_report(sd):
    """q = 0 (a landmark at the robot's position: NaN like NumPy's 0/0, src/replay_no_ros.py:466-469) comes back through the
    fused call with the sticky flag set; a wrong size is refused before anything is enqueued."""
    mean = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 1.0])
    with sd.EkfSlam(7) as f:
        f.set_option("small_state", 1)
        f.set_state(mean, np.eye(7))
        lib = sd.load_library()
        out_mu, out_P = np.empty(9), np.empty((9, 9))
        one, z = np.zeros(1), np.zeros(1)
        rc = lib.ekf_step_fetch(f._h, one.ctypes.data_as(C.POINTER(C.c_double)), z.ctypes.data_as(C.POINTER(C.c_double)),
                                None, None, None, np.zeros(1, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int)), 0, 0,
                                out_mu.ctypes.data_as(C.POINTER(C.c_double)), out_P.ctypes.data_as(C.POINTER(C.c_double)), 9)
        assert rc != 0 and small_launches(sd, f) == 0
        mu, P = f.step_state(0.0, 0.0, [], [], [])            # nothing observed, no motion: the pose stays where landmark 0 is
        om, oP = orc.ekf_step_dense(mean, np.eye(7), 0.0, 0.0, [], [], [], orc.EkfConfig())
        close(mu, om)
        close(P, oP)
        assert np.array_equal(mu[:2], mean[:2]) and f.flags() == 0
        mu, P = f.step_state(0.0, 0.0, [0], [0.1], [0.0])     # the landmark coincides with the pose: q == 0
        assert not np.isfinite(mu).all()
        assert f.flags() & 1
        assert fused_fetches(sd, f) == 2


@pytest.mark.parametrize("N,m", [(12, 3), (38, 8)])
def test_polled_hand_over_carries_an_integrity_trailer(sd, N, m):
    """ekf_step_fetch trusts the state it finds in pinned memory as soon as the sequence word arrives, while the launch is
    still running (that rests on the ordering of the kernel's posted writes).  Every hand-over therefore carries a
    second copy of the sequence number written by another wave and an XOR checksum of the payload; with `fetch_verify` the
    host checks both.  200 calls: every one answered by the step's own launch, none retried, each equal to a plain download."""
    lib = sd.load_library()
    s = orc.synthetic_stream(N, 200, m, 70)
    with sd.EkfSlam(3 + 2 * N) as f:
        f.set_option("small_state", 1)
        f.set_option("fetch_verify", 1)
        f.set_state_diag(s[0], s[1])
        for k in range(200):
            mu, P = f.step_state(s[2][k], s[3][k], s[4][k], s[5][k], s[6][k])
            if k % 20 == 0:
                dm, dP = f.state()
                assert np.array_equal(mu, dm) and np.array_equal(P, dP)
        assert fused_fetches(sd, f) == 200 and lib.ekf_debug_fetch_retries(f._h) == 0 and f.flags() == 0
    om, oP = s[0].copy(), np.diag(s[1])
    for k in range(200):
        om, oP = orc.ekf_step_dense(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], orc.EkfConfig())
    close(mu, om)
    close(P, oP)


def test_a_bank_that_overfills_the_chip_takes_the_throughput_kernel_bit_identically(sd):
    """More than three trajectories per CU: the small-state kernel runs in its 128-VGPR form, four workgroups resident per CU
    (k_small_stream_occ) -- the same instructions on the data, so sampled trajectories equal, bit for bit, the same
    trajectories run in a bank of two (the latency form), and the oracle."""
    N, steps, B = 12, 6, 800
    base = [orc.synthetic_stream(N, steps, 5, 60 + t) for t in range(8)]
    cols = [np.ascontiguousarray(np.stack([base[b % 8][i] for b in range(B)], 1)) for i in (2, 3, 4, 5, 6)]
    with sd.EkfSlam(3 + 2 * N, batch=B) as f:
        f.set_option("small_state", 1)
        for b in range(B):
            f.set_state_diag(base[b % 8][0], base[b % 8][1], b)
        f.stream_upload(*cols)
        f.stream_run(0, 3)                                  # one launch for three steps ...
        for k in range(3, steps):                           # ... then online steps
            f.step(cols[0][k], cols[1][k], list(cols[2][k]), list(cols[3][k]), list(cols[4][k]))
        big = {b: f.state(b) for b in (0, 5, 333, 799)}
        assert small_launches(sd, f) == 1 + (steps - 3) and not any(f.flags(b) for b in big)
    for b, (mu, P) in big.items():
        s = base[b % 8]
        with sd.EkfSlam(3 + 2 * N, batch=2) as g:
            g.set_option("small_state", 1)
            for t in range(2):
                g.set_state_diag(s[0], s[1], t)
            g.stream_upload(*[np.ascontiguousarray(np.stack([s[i], s[i]], 1)) for i in (2, 3, 4, 5, 6)])
            g.stream_run(0, 3)
            for k in range(3, steps):
                g.step([s[2][k]] * 2, [s[3][k]] * 2, [s[4][k]] * 2, [s[5][k]] * 2, [s[6][k]] * 2)
            mu2, P2 = g.state(1)
        assert np.array_equal(mu, mu2) and np.array_equal(P, P2), b
        om, oP = s[0].copy(), np.diag(s[1])
        for k in range(steps):
            om, oP = orc.ekf_step_dense(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], orc.EkfConfig())
        close(mu, om)
        close(P, oP)


@pytest.mark.parametrize("n_lm,n_cap", [(45, 45), (50, 64), (64, 64)])
def test_banks_take_the_small_path_up_to_64_landmarks(sd, n_lm, n_cap):
    """One trajectory of more than 38 landmarks is faster on the general kernels (latency); a bank that fills the chip is not:
    from 128 trajectories on a handle of up to n_max = 131 (64 landmarks) runs the small-state kernel, one workgroup per
    trajectory with up to 137 KB of LDS (7 or 9 column tiles).  Uploaded stream, online steps with 0 .. 20 observations, the
    fused step + state, a dense start: sampled trajectories against the oracle and against the same bank on the general kernels."""
    rng = np.random.default_rng(5 + n_lm)
    B, steps = 130, 6
    cfg = orc.EkfConfig()
    base = [orc.synthetic_stream(n_lm, steps, 8, 80 + t) for t in range(6)]
    n = 3 + 2 * n_lm
    A = rng.normal(size=(n, 4)) * 0.2
    dense0 = A @ A.T + np.diag(rng.uniform(0.5, 2.0, n))
    cols = [np.ascontiguousarray(np.stack([base[b % 6][i] for b in range(B)], 1)) for i in (2, 3, 4, 5, 6)]
    extra = []                                               # online steps behind the stream: 0, 3 and 20 observations
    for m in (0, 3, 20):
        idx = rng.choice(n_lm, size=m, replace=False).astype(np.int32)
        extra.append((0.004, 0.03, idx, rng.uniform(0.4, 1.5, m), rng.uniform(-0.5, 0.5, m)))
    out = {}
    with sd.EkfSlam(3 + 2 * n_cap, batch=1) as one:
        one.set_option("small_state", 1)
        one.set_state_diag(base[0][0], base[0][1])
        one.step(base[0][2][0], base[0][3][0], base[0][4][0], base[0][5][0], base[0][6][0])
        assert small_launches(sd, one) == 0                  # a single trajectory of this size: the general kernels
    for small in (1, 0):
        with sd.EkfSlam(3 + 2 * n_cap, batch=B) as f:
            f.set_option("small_state", small)
            for b in range(B):
                if b == 7:
                    f.set_state(base[b % 6][0], dense0, b)
                else:
                    f.set_state_diag(base[b % 6][0], base[b % 6][1], b)
            f.stream_upload(*cols)
            f.stream_run(0, steps)
            for k, (lin, ang, idx, zr, zb) in enumerate(extra):
                args = ([lin] * B, [ang] * B, [idx] * B, [zr] * B, [zb] * B)
                if k == 1:
                    mu_f, P_f = f.step_state(*args, b=7)
                else:
                    f.step(*args)
            out[small] = {b: f.state(b) for b in (0, 7, 64, 129)}
            assert np.array_equal(out[small][7][0], mu_f) or k == 2          # (the fused fetch returned the state of step 1)
            assert not any(f.flags(b) for b in out[small])
            assert small_launches(sd, f) == (1 + 4 if small else 0)          # the stream, then 1 + 1 + 2 launches (20 > 16 landmarks)
    for b in (0, 7, 64, 129):
        s = base[b % 6]
        om, oP = s[0].copy(), (dense0.copy() if b == 7 else np.diag(s[1]))
        for k in range(steps):
            om, oP = orc.ekf_step_dense(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], cfg)
        for lin, ang, idx, zr, zb in extra:
            om, oP = orc.ekf_step_dense(om, oP, lin, ang, idx, zr, zb, cfg)
        close(out[1][b][0], om)
        close(out[1][b][1], oP)
        close(out[1][b][0], out[0][b][0], 1e-10)
        close(out[1][b][1], out[0][b][1], 1e-10)
        assert np.array_equal(out[1][b][1], out[1][b][1].T)
