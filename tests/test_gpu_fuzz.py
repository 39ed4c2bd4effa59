"""Randomised mixed-API runs on the GPU against the reference-shaped dense oracle.

Every targeted test drives one path; a user's program mixes them.  Here a seeded random sequence of calls -- online steps
with 0..10 observations, predictions and updates on their own, augmentation, uploaded streams of random length (fused
cadences, look-ahead), explicit flushes, general dense propagations, downloads in between, option changes (pass cadence,
pass kernel, fused cadence / step, look-ahead) -- runs on a small bank of trajectories, and the state of every trajectory
is compared with `oracle.ekf_step_dense` & co. (pinned to the reference's golden vectors) after each download.  What this
exercises is the handle's bookkeeping around the kernels: pending ranks and their flush points, the double-buffered mean
and pose noise, the active bound across uploads / dense products / growing states, cadence planning with a pre-solved
cadence in flight, streams uploaded before the state changed.
Reference: src/replay_no_ros.py:341-360 (augmentation), :368-430 (prediction), :436-480 (sequential update).
"""
import numpy as np
import pytest

from oracle import ekf_oracle as orc

pytestmark = pytest.mark.gpu

REL_TOL = 1e-6
TIGHT = 1e-8          # (long random sequences from 1e4 initial variances: rounding grows a little beyond the 1e-9 of short runs)


@pytest.fixture(scope="module")
def sd():
    import slam_duckietown_amd as sd
    sd.load_library()
    return sd


def close(a, b, what):
    r = orc.rel_fro(a, b)
    assert r < REL_TOL, f"{what}: rel Frobenius {r:.3e} exceeds the 1e-6 bar"
    assert r < TIGHT, f"{what}: rel Frobenius {r:.3e} exceeds the expected {TIGHT:g}"


class Model:
    """The oracle's view of one trajectory."""

    def __init__(self, rng, n_lm, cfg):
        self.cfg = cfg
        self.truth = np.stack([rng.uniform(-1.0, 1.0, n_lm), rng.uniform(-0.8, 1.2, n_lm)], 1)
        self.mean = np.concatenate([[0.0, 0.0, 0.0], (self.truth + rng.normal(0, 0.05, self.truth.shape)).ravel()])
        self.diag = np.concatenate([[0.1, 0.1, 0.1], np.full(2 * n_lm, 1.0e4)])
        self.cov = np.diag(self.diag)
        self.pose = np.zeros(3)

    @property
    def n_lm(self):
        return (len(self.mean) - 3) // 2

    def observe(self, rng, m):
        """m distinct landmarks (fewer if the map is smaller), measured from the model's own pose estimate + noise."""
        m = min(m, self.n_lm)
        idx = rng.choice(self.n_lm, size=m, replace=False).astype(np.int32)
        lx, ly = self.mean[3 + 2 * idx], self.mean[4 + 2 * idx]
        dx, dy = lx - self.mean[0], ly - self.mean[1]
        zr = np.hypot(dx, dy) + rng.normal(0, 0.02, m)
        zb = np.arctan2(dy, dx) - self.mean[2] + rng.normal(0, 0.02, m)
        return idx, np.maximum(zr, 0.05), zb

    def grow(self, xy):
        k = len(xy)
        n = len(self.mean)
        self.mean = np.concatenate([self.mean, np.asarray(xy).reshape(-1)])
        cov = np.zeros((n + 2 * k, n + 2 * k))
        cov[:n, :n] = self.cov
        cov[np.arange(n, n + 2 * k), np.arange(n, n + 2 * k)] = self.cfg.landmark_init_var
        self.cov = cov


@pytest.mark.parametrize("seed,n_lm,batch,small", [(0, 30, 1, 0), (1, 70, 2, 0), (2, 140, 3, 0), (3, 260, 1, 0), (4, 45, 2, 0),
                                                   (5, 400, 1, 0), (6, 20, 2, 1), (7, 26, 3, 1), (8, 12, 1, 1)])
def test_random_mixed_api_sequences_against_the_oracle(sd, seed, n_lm, batch, small):
    rng = np.random.default_rng(1000 + seed)
    cfg = orc.EkfConfig()
    models = [Model(rng, n_lm, cfg) for _ in range(batch)]
    cap = 3 + 2 * (n_lm + 12)
    ops_done = []
    with sd.EkfSlam(cap, batch=batch) as f:
        f.set_option("small_state", small)           # (small: the whole state in LDS, csrc/ekf_small.hip; needs n_max <= 79)
        for b, mdl in enumerate(models):
            if rng.random() < 0.5:
                f.set_state_diag(mdl.mean, mdl.diag, b)
            else:
                f.set_state(mdl.mean, mdl.cov, b)
        n_ops = 45 if n_lm <= 300 else 25

        def check(b, what):
            mu, P = f.state(b)
            assert f.flags(b) == 0, (what, ops_done[-6:])
            assert len(mu) == len(models[b].mean)
            assert np.array_equal(P, P.T)
            close(mu, models[b].mean, f"{what}: mean of trajectory {b} after {ops_done[-6:]}")
            close(P, models[b].cov, f"{what}: covariance of trajectory {b} after {ops_done[-6:]}")

        for it in range(n_ops):
            op = rng.choice(["step", "step", "step", "predict", "update", "grow", "stream", "stream", "flush", "dense", "option",
                             "download"])
            lin = rng.uniform(0.002, 0.02, batch)
            ang = np.where(rng.random(batch) < 0.3, rng.uniform(-0.008, 0.008, batch), rng.uniform(-0.3, 0.3, batch))
            if op == "step":
                m = int(rng.integers(0, 11))
                obs = [mdl.observe(rng, m) for mdl in models]
                f.step(lin, ang, [o[0] for o in obs], [o[1] for o in obs], [o[2] for o in obs])
                for b, mdl in enumerate(models):
                    mdl.mean, mdl.cov = orc.ekf_step_dense(mdl.mean, mdl.cov, lin[b], ang[b], *obs[b], cfg)
            elif op == "predict":
                f.predict(lin, ang)
                for b, mdl in enumerate(models):
                    mdl.mean, mdl.cov = orc.predict_dense(mdl.mean, mdl.cov, lin[b], ang[b], cfg)
            elif op == "update":
                m = int(rng.integers(1, 20))                     # more than 16: split into passes by the library
                obs = [mdl.observe(rng, m) for mdl in models]
                f.update([o[0] for o in obs], [o[1] for o in obs], [o[2] for o in obs])
                for b, mdl in enumerate(models):
                    mdl.mean, mdl.cov = orc.update_dense(mdl.mean, mdl.cov, *obs[b], cfg)
            elif op == "grow":
                b = int(rng.integers(0, batch))
                k = int(rng.integers(1, 4))
                if models[b].n_lm + k <= n_lm + 12:
                    xy = rng.uniform(-1.0, 1.0, (k, 2))
                    f.add_landmarks(xy, b)
                    models[b].grow(xy)
            elif op == "stream":
                steps = int(rng.integers(1, 14))
                mcap = int(rng.choice([1, 2, 4, 8, 8, 8]))
                idx = np.zeros((steps, batch, mcap), dtype=np.int32)
                zr, zb = np.zeros((steps, batch, mcap)), np.zeros((steps, batch, mcap))
                ms = np.zeros((steps, batch), dtype=np.int32)
                lins, angs = rng.uniform(0.002, 0.02, (steps, batch)), rng.uniform(-0.2, 0.2, (steps, batch))
                angs[rng.random((steps, batch)) < 0.2] = 0.004
                for k in range(steps):
                    for b, mdl in enumerate(models):
                        m = int(rng.integers(0 if rng.random() < 0.2 else 1, mcap + 1))
                        o = mdl.observe(rng, m)
                        m = len(o[0])
                        ms[k, b] = m
                        idx[k, b, :m], zr[k, b, :m], zb[k, b, :m] = o
                        mdl.mean, mdl.cov = orc.ekf_step_dense(mdl.mean, mdl.cov, lins[k, b], angs[k, b], o[0], o[1], o[2], cfg)
                f.stream_upload(lins, angs, idx, zr, zb, ms)
                cut = int(rng.integers(0, steps + 1))
                f.stream_run(0, cut)                              # in two pieces: cadences end at the piece boundary
                if rng.random() < 0.3 and cut:
                    f.mean(int(rng.integers(0, batch)))           # a blocking read in between
                f.stream_run(cut, steps - cut)
            elif op == "flush":
                f.flush()
            elif op == "dense" and n_lm <= 150:
                b = int(rng.integers(0, batch))
                n = len(models[b].mean)
                F = np.eye(n) + rng.normal(size=(n, n)) * (0.05 / np.sqrt(n))
                A = rng.normal(size=(n, 3)) * 0.02
                Q = A @ A.T + np.diag(rng.uniform(1e-4, 1e-3, n))
                f.predict_dense(F, Q, b)
                P = F @ models[b].cov @ F.T + Q
                models[b].cov = np.triu(P) + np.triu(P, 1).T      # (the device keeps the upper triangle)
            elif op == "option":
                name, value = [("flush_every", int(rng.integers(0, 6))), ("pass_kernel", int(rng.choice([-1, 0, 2]))),
                               ("fused_cadence", int(rng.integers(0, 2))), ("fused_step", int(rng.integers(0, 2))),
                               ("lookahead", int(rng.integers(0, 2))), ("rank_limit", int(rng.choice([16, 32, 48, 80]))),
                               ("active_bound", int(rng.integers(0, 2))), ("pass_streaming", int(rng.choice([-1, 0, 1])))][
                    int(rng.integers(0, 8))]
                f.set_option(name, value)
                op = f"{name}={value}"
            ops_done.append(op)
            if op == "download" or rng.random() < 0.25:
                check(int(rng.integers(0, batch)), f"op {it}")
        for b in range(batch):
            check(b, "end")
        if small:                                             # (not vacuous: the small-state kernel is what ran)
            import ctypes as C
            lib = sd.load_library()
            assert lib.ekf_debug_small_launches(f._h) > 10
