"""GPU parity of the fused cadence (slam-duckietown_amd/csrc/ekf_cadence.hip): uploaded streams run everything between
two covariance passes as one solve launch + one panel launch.  Round 5: cadences are PACKED -- a trajectory's next 40
landmark updates whatever steps they belong to (a step may be cut by a pass), every trajectory on its own cursor.

The recurrences are those of the per-step kernels (src/replay_no_ros.py:368-480: motion model, P <- G P G^T + R, per
landmark H, S, K, mean and covariance update) in a different summation order -- the effect of the cadence's earlier
ranks is carried in registers instead of re-read from V and W -- so the two paths agree to rounding, not bit for bit.
Stated tolerance: 1e-11 relative Frobenius between the paths in these tests (measured: 1e-15 .. 1e-13 from dense,
well-conditioned starts; up to 1.2e-12 from the block-diagonal start, whose 1e4 landmark variances against a 0.49
measurement noise amplify a last-bit difference by four orders of magnitude; the header guarantees 1e-10), and the usual
1e-9 / 1e-6 against the oracle (the
reference-shaped dense NumPy path, pinned to the reference's golden vectors).
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import ekf_oracle as orc
from tests import golden_util as gu

pytestmark = pytest.mark.gpu

PATH_TOL = 1e-11
TIGHT = 1e-9


@pytest.fixture(scope="module")
def sd():
    import slam_duckietown_amd as sd
    sd.load_library()
    return sd


def cadences(sd, f):
    lib = sd.load_library()
    a, b = C.c_long(), C.c_long()
    assert lib.ekf_debug_cadences(f._h, C.byref(a), C.byref(b)) == 0
    return a.value, b.value


def cadences_needed(m_col, slots=40, steps_cap=40):
    """Cadences one trajectory's landmark counts need (the greedy of csrc/ekf_host_plan.h::plan_cadences, restated): whole
    steps while they fit the slots left, then a cut step fills them."""
    t, j, c = 0, 0, 0
    while t < len(m_col):
        c += 1
        used = ns = 0
        while t < len(m_col) and ns < steps_cap:
            left = int(m_col[t]) - j
            if used + left <= slots:
                used, ns, t, j = used + left, ns + 1, t + 1, 0
                continue
            if slots - used > 0:
                j += slots - used
            break
    return c


def dense_start(n, seed, rank=6):
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(n, rank)) * 0.3
    P = A @ A.T
    P[np.arange(n), np.arange(n)] += rng.uniform(0.5, 2.0, n)
    return P


def run_stream(sd, n, B, starts, means, lin, ang, idx, zr, zb, m=None, options=(), diag=False, cfg=None):
    """One handle, the whole stream; returns ([(mean, cov)] per trajectory, fused cadences, steps they covered)."""
    with sd.EkfSlam(n, batch=B, config=cfg) as f:
        for name, value in options:
            f.set_option(name, value)
        for b in range(B):
            if diag:
                f.set_state_diag(means[b], starts[b], b)
            else:
                f.set_state(means[b], starts[b], b)
        f.run_stream(lin, ang, idx, zr, zb, m)
        out = [f.state(b) for b in range(B)]
        assert [f.flags(b) for b in range(B)] == [0] * B
        c = cadences(sd, f)
    return out, c


def stack(streams, i):
    return np.stack([s[i] for s in streams], axis=1)


@pytest.mark.parametrize("N,B,m,steps", [(60, 1, 8, 12), (300, 3, 8, 11), (300, 2, 1, 85), (200, 2, 2, 43),
                                         (257, 2, 4, 23), (120, 2, 16, 5), (700, 9, 8, 10),
                                         (150, 2, 3, 30), (150, 2, 5, 17), (100, 2, 6, 14), (100, 1, 7, 12), (120, 2, 9, 9),
                                         (150, 2, 10, 9), (150, 2, 11, 8), (150, 2, 12, 7), (130, 3, 13, 7), (130, 1, 15, 6)])
def test_fused_cadence_equals_the_per_step_path_and_the_oracle(sd, N, B, m, steps):
    """Every landmark count per step from 1 to 16 that matters (powers of two -- the round-3 slot sizes -- and the counts
    between them, where 40 slots end inside a step and the pass cuts it), whole cadences plus a tail, dense starting
    covariances, latency and throughput shapes of the panel launch: fused == per-step to PATH_TOL (1e-10 guaranteed by the header, 1e-13 .. 1e-12 measured), == oracle to 1e-9."""
    n = 3 + 2 * N
    streams = [orc.synthetic_stream(N, steps, m, 300 + t) for t in range(B)]
    starts = [dense_start(n, 400 + t) for t in range(B)]
    means = [s[0] for s in streams]
    args = (stack(streams, 2), stack(streams, 3), stack(streams, 4), stack(streams, 5), stack(streams, 6))
    fused, (nc, ns) = run_stream(sd, n, B, starts, means, *args, options=[("active_bound", 0)])
    plain, (pc, _) = run_stream(sd, n, B, starts, means, *args, options=[("active_bound", 0), ("fused_cadence", 0)])
    assert pc == 0 and nc == -(-steps * m // 40) and ns == steps      # exactly 2 ranks per landmark update: 40 updates per pass
    for b in range(B):
        assert orc.rel_fro(fused[b][0], plain[b][0]) < PATH_TOL
        assert orc.rel_fro(fused[b][1], plain[b][1]) < PATH_TOL
        assert np.array_equal(fused[b][1], fused[b][1].T)
    cfg = orc.EkfConfig()
    s = streams[B - 1]
    om, oP = s[0].copy(), starts[B - 1].copy()
    step = orc.ekf_step_dense if n < 1000 else orc.ekf_step_structured    # (the O(n^2) form where dense takes seconds per step)
    for k in range(steps):
        om, oP = step(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], cfg)
    assert orc.rel_fro(fused[B - 1][0], om) < TIGHT and orc.rel_fro(fused[B - 1][1], oP) < TIGHT


def test_ragged_observations_repeated_landmarks_and_empty_steps(sd):
    """What real windows look like: per trajectory and step 0..8 observations (an empty step inside a cadence, a
    trajectory that sees nothing at all in one cadence), and the SAME landmarks observed in consecutive steps (two
    slots of the cadence then hold one state index).  Against the per-step path and the oracle."""
    N, B, steps, M = 90, 3, 17, 8
    n = 3 + 2 * N
    rng = np.random.default_rng(5)
    world = [orc.synthetic_world(N, 20 + t) for t in range(B)]
    cfg = orc.EkfConfig()
    lin = np.full((steps, B), 0.004)
    ang = np.where(np.arange(steps)[:, None] % 4 == 3, 0.005, 0.02) * np.ones((1, B))
    idx = np.zeros((steps, B, M), dtype=np.int32)
    zr = np.zeros((steps, B, M))
    zb = np.zeros((steps, B, M))
    m = np.zeros((steps, B), dtype=np.int32)
    pose = [np.zeros(3) for _ in range(B)]
    for k in range(steps):
        for b in range(B):
            pose[b], _ = orc.motion_model(pose[b], lin[k, b], ang[k, b], cfg)
            if b == 2 and 5 <= k < 10:
                mb = 0                                    # trajectory 2 sees nothing during the whole second cadence
            elif k % 6 == 2:
                mb = 0                                    # an empty step inside a cadence
            else:
                mb = int(rng.integers(1, M + 1))
            # landmarks come from a small pool: repeats across the steps of a cadence are the rule
            vis = rng.choice(12, size=mb, replace=False) + (0 if k < 9 else 30)
            d = world[b][1][vis] - pose[b][0:2]
            cth, sth = np.cos(pose[b][2]), np.sin(pose[b][2])
            xr = cth * d[:, 0] + sth * d[:, 1] + rng.normal(0, 0.01, mb)
            yr = -sth * d[:, 0] + cth * d[:, 1] + rng.normal(0, 0.01, mb)
            m[k, b] = mb
            idx[k, b, :mb] = vis
            zr[k, b, :mb] = np.sqrt(xr ** 2 + yr ** 2)
            zb[k, b, :mb] = np.arctan2(yr, xr)
    # every step has some trajectory with 5..8 observations, so that all steps take the 8-landmark slot size
    m[:, 0] = np.maximum(m[:, 0], 5)
    for k in range(steps):
        if not zr[k, 0, :m[k, 0]].all():                  # (rows whose count was raised: fill them in)
            vis = rng.choice(12, size=m[k, 0], replace=False) + 50
            idx[k, 0, :m[k, 0]] = vis
            zr[k, 0, :m[k, 0]] = 0.3 + 0.05 * np.arange(m[k, 0])
            zb[k, 0, :m[k, 0]] = 0.1 * np.arange(m[k, 0]) - 0.3
    means = [w[2] for w in world]
    starts = [dense_start(n, 500 + t) for t in range(B)]
    fused, (nc, ns) = run_stream(sd, n, B, starts, means, lin, ang, idx, zr, zb, m, options=[("active_bound", 0)])
    plain, _ = run_stream(sd, n, B, starts, means, lin, ang, idx, zr, zb, m,
                          options=[("active_bound", 0), ("fused_cadence", 0)])
    assert nc == max(cadences_needed(m[:, b]) for b in range(B)) and ns == 17   # every trajectory on its own cursor
    for b in range(B):
        assert orc.rel_fro(fused[b][0], plain[b][0]) < PATH_TOL and orc.rel_fro(fused[b][1], plain[b][1]) < PATH_TOL
        om, oP = means[b].copy(), starts[b].copy()
        for k in range(steps):
            mb = m[k, b]
            om, oP = orc.ekf_step_dense(om, oP, lin[k, b], ang[k, b], idx[k, b, :mb], zr[k, b, :mb], zb[k, b, :mb], cfg)
        assert orc.rel_fro(fused[b][0], om) < TIGHT and orc.rel_fro(fused[b][1], oP) < TIGHT


def test_block_diagonal_start_with_the_active_bound(sd):
    """The benchmark's god-mode style start (P0 block diagonal) with the active bound ON: state indices
    beyond the highest landmark seen so far are skipped by the cadence's panel launch and by the pass; trajectories of
    different sizes in one batch."""
    N, B, steps, m = 260, 3, 13, 8
    streams = [orc.synthetic_stream(N, steps, m, 600 + t) for t in range(B)]
    sizes = [3 + 2 * N, 3 + 2 * 100, 3 + 2 * 201]
    means = [s[0][:sizes[b]] for b, s in enumerate(streams)]
    diags = [s[1][:sizes[b]] for b, s in enumerate(streams)]
    idx = np.stack([(s[4] * 3 + 1) % ((sizes[b] - 3) // 2 // 2) for b, s in enumerate(streams)], axis=1)
    # (the remapped indices may repeat inside a step: keep the first of each)
    mm = np.zeros((steps, B), dtype=np.int32)
    for k in range(steps):
        for b in range(B):
            _, first = np.unique(idx[k, b], return_index=True)
            keep = np.sort(first)
            mm[k, b] = len(keep)
            idx[k, b, :len(keep)] = idx[k, b, keep]
    args = (stack(streams, 2), stack(streams, 3), idx, stack(streams, 5), stack(streams, 6), mm)
    fused, (nc, _) = run_stream(sd, 3 + 2 * N, B, diags, means, *args, diag=True)
    plain, _ = run_stream(sd, 3 + 2 * N, B, diags, means, *args, diag=True, options=[("fused_cadence", 0)])
    dense, _ = run_stream(sd, 3 + 2 * N, B, diags, means, *args, diag=True, options=[("active_bound", 0)])
    assert nc >= 2
    for b in range(B):
        assert fused[b][0].shape == (sizes[b],)
        for other in (plain, dense):
            assert orc.rel_fro(fused[b][0], other[b][0]) < PATH_TOL and orc.rel_fro(fused[b][1], other[b][1]) < PATH_TOL
    cfg = orc.EkfConfig()
    om, oP = means[1].copy(), np.diag(diags[1])
    for k in range(steps):
        om, oP = orc.ekf_step_dense(om, oP, streams[1][2][k], streams[1][3][k], idx[k, 1, :mm[k, 1]],
                                    streams[1][5][k][:mm[k, 1]], streams[1][6][k][:mm[k, 1]], cfg)
    assert orc.rel_fro(fused[1][0], om) < TIGHT and orc.rel_fro(fused[1][1], oP) < TIGHT


@pytest.mark.parametrize("options,expect", [([("rank_limit", 48)], (5, 13)), ([("flush_every", 2)], (7, 13)),
                                            ([("flush_every", 1)], (13, 13)), ([("rank_limit", 16)], (13, 13)),
                                            ([("rank_limit", 22)], (10, 13))])
def test_cadence_follows_the_pass_cadence_options(sd, options, expect):
    """`rank_limit` / `flush_every` shorten the cadence (24, 8 or 11 landmark updates -- the last cuts every step --, 2 steps
    or 1 step here); the result does not depend on it beyond rounding."""
    N, B, steps, m = 150, 2, 13, 8
    n = 3 + 2 * N
    streams = [orc.synthetic_stream(N, steps, m, 700 + t) for t in range(B)]
    starts = [dense_start(n, 800 + t) for t in range(B)]
    means = [s[0] for s in streams]
    args = (stack(streams, 2), stack(streams, 3), stack(streams, 4), stack(streams, 5), stack(streams, 6))
    ref, _ = run_stream(sd, n, B, starts, means, *args, options=[("active_bound", 0), ("fused_cadence", 0)])
    got, c = run_stream(sd, n, B, starts, means, *args, options=[("active_bound", 0)] + options)
    assert c == expect
    for b in range(B):
        assert orc.rel_fro(got[b][0], ref[b][0]) < PATH_TOL and orc.rel_fro(got[b][1], ref[b][1]) < PATH_TOL


@pytest.mark.parametrize("flags", [dict(disable_motion_model=True), dict(enable_circular_interpolation=False),
                                   dict(enable_measurement_model=False)])
def test_config_flags_through_the_cadence(sd, flags):
    """The reference's module flags (src/replay_no_ros.py:18-19, :28) through the fused path; with the measurement
    model off a step observes nothing: the 11 predictions are ONE cadence that appends no rank and needs no pass (the
    panel launch applies the motion noise itself)."""
    N, B, steps, m = 80, 2, 11, 8
    n = 3 + 2 * N
    streams = [orc.synthetic_stream(N, steps, m, 900 + t) for t in range(B)]
    starts = [dense_start(n, 950 + t) for t in range(B)]
    means = [s[0] for s in streams]
    args = (stack(streams, 2), stack(streams, 3), stack(streams, 4), stack(streams, 5), stack(streams, 6))
    got, (nc, _) = run_stream(sd, n, B, starts, means, *args, options=[("active_bound", 0)], cfg=sd.EkfConfig(**flags))
    assert nc == (1 if "enable_measurement_model" in flags else 3)
    ocfg = orc.EkfConfig(**flags)
    for b in range(B):
        s = streams[b]
        om, oP = s[0].copy(), starts[b].copy()
        for k in range(steps):
            om, oP = orc.ekf_step_dense(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], ocfg)
        assert orc.rel_fro(got[b][0], om) < TIGHT and orc.rel_fro(got[b][1], oP) < TIGHT


def test_all_three_shapes_of_the_panel_launch_agree(sd):
    """The panel launch of a cadence has three shapes by the number of state indices in the launch: rows of the panel
    split over the four waves of a workgroup (k_panels_cad_ks: up to 512 waves of state indices), one wave per workgroup
    (up to 1024), four independent waves per workgroup (beyond).  N = 1300 with 3, 16 and 30 trajectories walks through
    all three; trajectory b of every batch runs the same stream from the same block-diagonal start, so the three must
    return the same state for it -- bit for bit: the forms perform the same operations in the same order."""
    N, steps, m = 1300, 7, 8
    n = 3 + 2 * N
    streams = [orc.synthetic_stream(N, steps, m, 1100 + t) for t in range(3)]
    res = {}
    for B in (3, 16, 30):
        pick = [streams[b % 3] for b in range(B)]
        args = (stack(pick, 2), stack(pick, 3), stack(pick, 4), stack(pick, 5), stack(pick, 6))
        out, (nc, ns) = run_stream(sd, n, B, [s[1] for s in pick], [s[0] for s in pick], *args, diag=True,
                                   options=[("active_bound", 0), ("lookahead", 0)])   # (the look-ahead is a matter of launch size too)
        assert (nc, ns) == (2, 7)
        res[B] = out
    for b in range(3):
        for B in (16, 30):
            assert np.array_equal(res[3][b][0], res[B][b][0]) and np.array_equal(res[3][b][1], res[B][b][1]), (B, b)
    plain, _ = run_stream(sd, n, 3, [s[1] for s in streams], [s[0] for s in streams], stack(streams, 2), stack(streams, 3),
                          stack(streams, 4), stack(streams, 5), stack(streams, 6), diag=True,
                          options=[("active_bound", 0), ("fused_cadence", 0)])
    for b in range(3):
        assert orc.rel_fro(res[3][b][0], plain[b][0]) < PATH_TOL and orc.rel_fro(res[3][b][1], plain[b][1]) < PATH_TOL


def test_forced_panel_shapes_agree_bit_for_bit(sd):
    """`panel_shape` forces a shape of the panel launch whatever its size (diagnostics): 1 the row-split latency form (one LDS
    broadcast per K row), 2 / 3 one / four independent waves per workgroup (round 6: the K rows of a down-date read sixteen at a
    time and handed to the FMAs by the fp64 DPP broadcast, v_fmac_f64_dpp row_newbcast).  Every shape performs the same fused
    operations in the same order on every entry: N = 1300 x 30 (the size takes shape 3) and N = 2100 x 3 (dense covariances;
    the size takes shape 1), every forced shape against the shape the size selects -- bit for bit."""
    for N, B, steps, m in ((1300, 30, 7, 8), (2100, 3, 9, 5)):
        n = 3 + 2 * N
        streams = [orc.synthetic_stream(N, steps, m, 1500 + t) for t in range(3)]
        pick = [streams[b % 3] for b in range(B)]
        args = (stack(pick, 2), stack(pick, 3), stack(pick, 4), stack(pick, 5), stack(pick, 6))
        starts = [s[1] for s in pick] if B > 3 else [dense_start(n, 1600 + t) for t in range(B)]   # (the small bank: dense covariances)
        res = {}
        for shape in (0, 1, 2, 3):
            out, (nc, ns) = run_stream(sd, n, B, starts, [s[0] for s in pick], *args, diag=B > 3,
                                       options=[("active_bound", 0), ("lookahead", 0), ("panel_shape", shape)])
            assert nc >= 1 and ns == steps
            res[shape] = out
        for shape in (1, 2, 3):
            for b in range(B):
                assert np.array_equal(res[0][b][0], res[shape][b][0]) and np.array_equal(res[0][b][1], res[shape][b][1]), (N, shape, b)


def test_pass_forms_w_from_v_bit_identical(sd):
    """`w_from_v` = 1 (the default): where a fused cadence's covariance pass follows its panel launch at once in the row-slab form, the panel
    launch writes V only (W is half of its stores) and the pass forms its W fragments from V and the records' S^-1 with the
    panel launch's own operations -- bit for bit the result of `w_from_v` = 0, the road taken counted.  N = 1300 x 30
    (block-diagonal starts, without and with the active bound), N = 2100 x 9 with ragged landmark counts (dense starts)."""
    lib = sd.load_library()
    for N, B, steps, hi, diag, bound in ((1300, 30, 9, 8, True, 0), (1300, 30, 9, 8, True, 1), (2100, 9, 8, 13, False, 0)):
        n = 3 + 2 * N
        means, lin, ang, idx, zr, zb, m = wandering_stream(N, B, steps, lambda k, b, rng: rng.integers(hi // 2, hi + 1), 6100 + N)
        if diag:
            starts = [np.concatenate([np.full(3, 0.1), np.full(2 * N, 1.0e4)]) for _ in range(B)]
        else:
            starts = [dense_start(n, 6200 + t) for t in range(B)]
        res = {}
        for wv in (0, 1):
            with sd.EkfSlam(n, batch=B) as f:
                f.set_option("active_bound", bound)
                f.set_option("w_from_v", wv)
                for b in range(B):
                    (f.set_state_diag if diag else f.set_state)(means[b], starts[b], b)
                f.run_stream(lin, ang, idx, zr, zb, m)
                res[wv] = [f.state(b) for b in range(B)]
                assert [f.flags(b) for b in range(B)] == [0] * B
                took = lib.ekf_debug_w_from_v(f._h)
                assert (took >= 1) if wv else (took == 0), (N, wv, took)
        for b in range(B):
            assert np.array_equal(res[0][b][0], res[1][b][0]) and np.array_equal(res[0][b][1], res[1][b][1]), (N, b)


def lookaheads(sd, f):
    lib = sd.load_library()
    return lib.ekf_debug_lookaheads(f._h)


def chained(sd, f):
    lib = sd.load_library()
    return lib.ekf_debug_chained(f._h)


# the three orders a run's cadences can be enqueued in (csrc/ekf_api.hip: enqueue_cadence): chained solves (round 6, default),
# the round-3 look-ahead, everything one behind the other
MODES = {"chain": (("lookahead", 1), ("chain", 1)), "lookahead": (("lookahead", 1), ("chain", 0)), "plain": (("lookahead", 0),)}


@pytest.mark.parametrize("N,B,m,steps", [(1250, 1, 8, 26), (700, 4, 8, 17), (900, 2, 1, 125), (1000, 2, 16, 9), (2000, 1, 8, 22)])
def test_lookahead_solve_beside_the_pass(sd, N, B, m, steps):
    """Small launches with at least ~48 MB of covariance (N = 1250 x 1, 700 x 4, 900 x 2, 1000 x 2, and BASELINE config 3,
    N = 2000 x 1): the solve of the next cadence runs beside the covariance pass of this one.  Round 6, CHAINED (the default):
    the solves follow one another on the handle's stream, each block formed by k_chain_cad from the previous cadence's records
    (T, S^-1, the pose rows of its transform) and P_base as it stood before that cadence, panel launch and pass on the second
    stream.  Round 3, look-ahead (`chain=0`): the block from P_base and the still pending ranks (k_gather_cad).  Both against
    the same stream with every solve behind its pass (`lookahead=0`) to PATH_TOL, against the oracle to 1e-9, the road taken
    asserted; dense starting covariances, so that the block carries every term."""
    n = 3 + 2 * N
    streams = [orc.synthetic_stream(N, steps, m, 1200 + t) for t in range(B)]
    starts = [dense_start(n, 1300 + t) for t in range(B)]
    args = (stack(streams, 2), stack(streams, 3), stack(streams, 4), stack(streams, 5), stack(streams, 6))
    res = {}
    for mode, opts in MODES.items():
        with sd.EkfSlam(n, batch=B) as f:
            f.set_option("active_bound", 0)
            for name, value in opts:
                f.set_option(name, value)
            for b in range(B):
                f.set_state(streams[b][0], starts[b], b)
            f.run_stream(*args)
            res[mode] = [f.state(b) for b in range(B)]
            assert [f.flags(b) for b in range(B)] == [0] * B
            want = (-(-steps * m // 40) - 1) if mode != "plain" else 0       # every cadence but the first is solved ahead
            assert lookaheads(sd, f) == want, (mode, lookaheads(sd, f), want)
            assert chained(sd, f) == (want if mode == "chain" else 0), (mode, chained(sd, f))
    for mode in ("chain", "lookahead"):
        for b in range(B):
            assert orc.rel_fro(res[mode][b][0], res["plain"][b][0]) < PATH_TOL, mode
            assert orc.rel_fro(res[mode][b][1], res["plain"][b][1]) < PATH_TOL, mode
    cfg = orc.EkfConfig()
    s = streams[0]
    om, oP = s[0].copy(), starts[0].copy()
    for k in range(steps):                             # (the O(n^2) form of the oracle: n is in the thousands here)
        om, oP = orc.ekf_step_structured(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], cfg)
    for mode in ("chain", "lookahead"):
        assert orc.rel_fro(res[mode][0][0], om) < TIGHT and orc.rel_fro(res[mode][0][1], oP) < TIGHT, mode


def test_chained_solves_options_agree(sd):
    """The pieces of the chained order one by one (N = 900 x 2, m ~ U{0..12}: cadences cut steps, trajectories at different
    steps): the cadence's inputs formed one cadence ahead (`pre_positions`) and the panel launch as its own gate
    (`panel_own_gate` = 1, against the default one-lane gate launch) change WHEN things are computed, not what: bit for bit.  The
    triangular-solve form of the chained panel launch (`panel_tform`, k_panels_cad_tf) against the replay form
    (k_panels_cad_ks): the same algebra in another order of summation, equal to PATH_TOL; every variant chained at every cadence."""
    N, B, steps = 900, 2, 16
    n = 3 + 2 * N
    means, lin, ang, idx, zr, zb, m = wandering_stream(N, B, steps, lambda k, b, rng: rng.integers(0, 13), 9100)
    starts = [dense_start(n, 9200 + t) for t in range(B)]
    res = {}
    for key, opts in {"default": (), "inline_positions": (("pre_positions", 0),), "gate_launch": (("panel_own_gate", 1),),
                      "replay_panel": (("panel_tform", 0),)}.items():
        with sd.EkfSlam(n, batch=B) as f:
            f.set_option("active_bound", 0)
            for name, value in opts:
                f.set_option(name, value)
            for b in range(B):
                f.set_state(means[b], starts[b], b)
            f.run_stream(lin, ang, idx, zr, zb, m)
            res[key] = [f.state(b) for b in range(B)]
            assert [f.flags(b) for b in range(B)] == [0] * B
            nc = cadences(sd, f)[0]
            assert nc >= 3 and chained(sd, f) == nc - 1
    for b in range(B):
        for key in ("inline_positions", "gate_launch"):
            assert np.array_equal(res[key][b][0], res["default"][b][0]) and np.array_equal(res[key][b][1], res["default"][b][1]), key
        assert orc.rel_fro(res["replay_panel"][b][0], res["default"][b][0]) < PATH_TOL
        assert orc.rel_fro(res["replay_panel"][b][1], res["default"][b][1]) < PATH_TOL


def test_lookahead_with_wandering_landmark_counts(sd):
    """The look-ahead with packed cadences whose steps are cut by the pass: the next cadence's block is gathered (k_gather_cad,
    from the plan's positions) while this cadence's ranks are pending, for trajectories that sit at different steps of the
    stream.  N = 1250 x 2, m ~ U{0..16}: against `lookahead` = 0 to rounding and against the O(n^2) oracle."""
    N, B, steps = 1250, 2, 14
    n = 3 + 2 * N
    means, lin, ang, idx, zr, zb, m = wandering_stream(N, B, steps, lambda k, b, rng: rng.integers(0, 17), 7100)
    starts = [dense_start(n, 7200 + t) for t in range(B)]
    res = {}
    for mode, opts in MODES.items():
        with sd.EkfSlam(n, batch=B) as f:
            f.set_option("active_bound", 0)
            for name, value in opts:
                f.set_option(name, value)
            for b in range(B):
                f.set_state(means[b], starts[b], b)
            f.run_stream(lin, ang, idx, zr, zb, m)
            res[mode] = [f.state(b) for b in range(B)]
            assert [f.flags(b) for b in range(B)] == [0] * B
            nc = cadences(sd, f)[0]
            assert nc == max(cadences_needed(m[:, b]) for b in range(B)) and nc >= 3
            assert lookaheads(sd, f) == (nc - 1 if mode != "plain" else 0)
            assert chained(sd, f) == (nc - 1 if mode == "chain" else 0)
    cfg = orc.EkfConfig()
    for b in range(B):
        om, oP = means[b].copy(), starts[b].copy()
        for k in range(steps):
            mb = m[k, b]
            om, oP = orc.ekf_step_structured(om, oP, lin[k, b], ang[k, b], idx[k, b, :mb], zr[k, b, :mb], zb[k, b, :mb], cfg)
        for mode in ("chain", "lookahead"):
            assert orc.rel_fro(res[mode][b][0], res["plain"][b][0]) < PATH_TOL, mode
            assert orc.rel_fro(res[mode][b][1], res["plain"][b][1]) < PATH_TOL, mode
            assert orc.rel_fro(res[mode][b][0], om) < TIGHT and orc.rel_fro(res[mode][b][1], oP) < TIGHT, mode


def test_lookahead_beside_the_row_slab_pass_on_static_shares(sd):
    """A few long trajectories (N = 8000 x 1) take the row-slab pass on equal static shares, on one workgroup per trajectory
    fewer than the chip has CUs, so that the next cadence's solve runs beside it as well.  Here at a size the oracle
    handles, the shares forced through `pass_workgroups`: against `lookahead=0` to rounding, against the oracle, and the
    launches really took that road (look-aheads counted, the last pass on a share table)."""
    lib = sd.load_library()
    N, B, m, steps = 700, 3, 8, 27
    n = 3 + 2 * N
    streams = [orc.synthetic_stream(N, steps, m, 2300 + t) for t in range(B)]
    starts = [dense_start(n, 2400 + t) for t in range(B)]
    args = (stack(streams, 2), stack(streams, 3), stack(streams, 4), stack(streams, 5), stack(streams, 6))
    res = {}
    for mode, opts in MODES.items():
        with sd.EkfSlam(n, batch=B) as f:
            for name, value in (("active_bound", 0), ("pass_streaming", 1), ("pass_kernel", 2), ("pass_workgroups", 8)) + opts:
                f.set_option(name, value)
            for b in range(B):
                f.set_state(streams[b][0], starts[b], b)
            f.run_stream(*args)
            f.flush()
            res[mode] = [f.state(b) for b in range(B)]
            assert [f.flags(b) for b in range(B)] == [0] * B
            assert lib.ekf_debug_last_pass_shares(f._h) >= 1
            assert (lookaheads(sd, f) >= 4) if mode != "plain" else (lookaheads(sd, f) == 0)
            assert chained(sd, f) == (lookaheads(sd, f) if mode == "chain" else 0)
    cfg = orc.EkfConfig()
    s = streams[1]
    om, oP = s[0].copy(), starts[1].copy()
    for k in range(steps):
        om, oP = orc.ekf_step_dense(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], cfg)
    for mode in ("chain", "lookahead"):
        for b in range(B):
            assert orc.rel_fro(res[mode][b][0], res["plain"][b][0]) < PATH_TOL, mode
            assert orc.rel_fro(res[mode][b][1], res["plain"][b][1]) < PATH_TOL, mode
        assert orc.rel_fro(res[mode][1][0], om) < TIGHT and orc.rel_fro(res[mode][1][1], oP) < TIGHT, mode


def test_stream_run_in_pieces_with_flushes_and_downloads_in_between(sd):
    """`stream_run(first, count)` called in pieces that cut cadences (and look-ahead chains) anywhere, with `flush()`,
    `mean()` and `covariance_block()` between them: every piece starts from whatever is pending (a cadence only forms
    where nothing is), and the result is the whole stream's to rounding.  At N = 1250 x 1 the look-ahead applies, so the
    second stream is exercised across API calls too."""
    N, B, m, steps = 1250, 1, 8, 41
    n = 3 + 2 * N
    s = orc.synthetic_stream(N, steps, m, 1500)
    start = dense_start(n, 1501)
    args = (s[2][:, None], s[3][:, None], s[4][:, None], s[5][:, None], s[6][:, None])
    whole, (nc, _) = run_stream(sd, n, B, [start], [s[0]], *args, options=[("active_bound", 0)])
    assert nc == 9                                        # 41 steps x 8 landmarks = 8 x 40 + 8
    with sd.EkfSlam(n) as f:
        f.set_option("active_bound", 0)
        f.set_state(s[0], start)
        f.stream_upload(*args)
        k = 0
        for i, count in enumerate((7, 1, 12, 3, 10, 8)):
            f.stream_run(k, count)
            k += count
            if i == 1:
                f.flush()
            elif i == 2:
                assert np.isfinite(f.mean()).all()
            elif i == 3:
                assert np.isfinite(f.covariance_block(0, 0, 3, 3)).all()
        assert k == steps
        mu, P = f.state()
        assert f.flags() == 0 and lookaheads(sd, f) >= 2
    assert orc.rel_fro(mu, whole[0][0]) < PATH_TOL and orc.rel_fro(P, whole[0][1]) < PATH_TOL


def test_short_pieces_with_run_end_flush_all_run_fused(sd):
    """a caller that drives `stream_run` in short pieces.  By default a piece's last cadence leaves its ranks
    pending, and the next piece runs the per-step kernels until the pass is due; with `run_end_flush` = 1 every piece ends with
    its covariance pass and EVERY step of every piece runs as a fused cadence -- counted -- with the whole stream's result
    to rounding either way."""
    N, B, m, steps = 150, 2, 3, 36
    n = 3 + 2 * N
    streams = [orc.synthetic_stream(N, steps, m, 8800 + t) for t in range(B)]
    starts = [dense_start(n, 8900 + t) for t in range(B)]
    args = (stack(streams, 2), stack(streams, 3), stack(streams, 4), stack(streams, 5), stack(streams, 6))
    whole, _ = run_stream(sd, n, B, starts, [s[0] for s in streams], *args, options=[("active_bound", 0)])
    covered = {}
    for ref in (0, 1):
        with sd.EkfSlam(n, batch=B) as f:
            f.set_option("active_bound", 0)
            f.set_option("small_state", 0)
            f.set_option("run_end_flush", ref)
            for b in range(B):
                f.set_state(streams[b][0], starts[b], b)
            f.stream_upload(*args)
            for k in range(0, steps, 4):                  # 12 landmark updates per piece
                f.stream_run(k, 4)
            out = [f.state(b) for b in range(B)]
            covered[ref] = cadences(sd, f)[1]
        for b in range(B):
            assert orc.rel_fro(out[b][0], whole[b][0]) < PATH_TOL and orc.rel_fro(out[b][1], whole[b][1]) < PATH_TOL
    assert covered[1] == steps and covered[0] < steps


def test_slot_size_changes_along_the_stream(sd):
    """The number of observations per step wanders (runs of 8, 3, 1, 16, 8, 2, 5 landmarks; the second trajectory anything
    up to that): until round 4 a cadence only took steps of one rank-slot size and the stream alternated between fused
    cadences and per-step kernels; packed cadences take everything.  Against the per-step path and the oracle."""
    N, B, steps = 200, 2, 46
    n = 3 + 2 * N
    rng = np.random.default_rng(77)
    sizes = [8] * 7 + [3] * 6 + [1] * 9 + [16] * 5 + [8] * 2 + [2] * 11 + [5] * 6     # runs of a size, of odd lengths
    assert len(sizes) == steps
    world = [orc.synthetic_world(N, 40 + t) for t in range(B)]
    cfg = orc.EkfConfig()
    M = 16
    lin = np.full((steps, B), 0.004)
    ang = np.full((steps, B), 0.02)
    idx = np.zeros((steps, B, M), dtype=np.int32)
    zr = np.zeros((steps, B, M))
    zb = np.zeros((steps, B, M))
    m = np.zeros((steps, B), dtype=np.int32)
    pose = [np.zeros(3) for _ in range(B)]
    for k in range(steps):
        for b in range(B):
            pose[b], _ = orc.motion_model(pose[b], lin[k, b], ang[k, b], cfg)
            mb = sizes[k] if b == 0 else int(rng.integers(0, sizes[k] + 1))   # trajectory 0 sets the step's slot size
            vis = rng.choice(N, size=mb, replace=False)
            d = world[b][1][vis] - pose[b][0:2]
            cth, sth = np.cos(pose[b][2]), np.sin(pose[b][2])
            xr = cth * d[:, 0] + sth * d[:, 1] + rng.normal(0, 0.01, mb)
            yr = -sth * d[:, 0] + cth * d[:, 1] + rng.normal(0, 0.01, mb)
            m[k, b] = mb
            idx[k, b, :mb] = vis
            zr[k, b, :mb] = np.sqrt(xr ** 2 + yr ** 2)
            zb[k, b, :mb] = np.arctan2(yr, xr)
    means = [w[2] for w in world]
    starts = [dense_start(n, 1600 + t) for t in range(B)]
    fused, (nc, ns) = run_stream(sd, n, B, starts, means, lin, ang, idx, zr, zb, m, options=[("active_bound", 0)])
    plain, _ = run_stream(sd, n, B, starts, means, lin, ang, idx, zr, zb, m,
                          options=[("active_bound", 0), ("fused_cadence", 0)])
    assert nc == max(cadences_needed(m[:, b]) for b in range(B)) and ns == steps   # all of it fused, 40 updates per pass
    for b in range(B):
        assert orc.rel_fro(fused[b][0], plain[b][0]) < PATH_TOL and orc.rel_fro(fused[b][1], plain[b][1]) < PATH_TOL
        om, oP = means[b].copy(), starts[b].copy()
        for k in range(steps):
            mb = m[k, b]
            om, oP = orc.ekf_step_dense(om, oP, lin[k, b], ang[k, b], idx[k, b, :mb], zr[k, b, :mb], zb[k, b, :mb], cfg)
        assert orc.rel_fro(fused[b][0], om) < TIGHT and orc.rel_fro(fused[b][1], oP) < TIGHT


def wandering_stream(N, B, steps, counts, seed):
    """Streams whose landmark count per step and trajectory is counts(k, b, rng), indices scattered (no order, no locality)."""
    rng = np.random.default_rng(seed)
    world = [orc.synthetic_world(N, seed + 1 + t) for t in range(B)]
    cfg = orc.EkfConfig()
    M = 16
    lin = np.full((steps, B), 0.004)
    ang = np.where(np.arange(steps)[:, None] % 7 == 6, 0.005, 0.02) * np.ones((1, B))
    idx = np.zeros((steps, B, M), dtype=np.int32)
    zr = np.zeros((steps, B, M))
    zb = np.zeros((steps, B, M))
    m = np.zeros((steps, B), dtype=np.int32)
    pose = [np.zeros(3) for _ in range(B)]
    for k in range(steps):
        for b in range(B):
            pose[b], _ = orc.motion_model(pose[b], lin[k, b], ang[k, b], cfg)
            mb = int(counts(k, b, rng))
            vis = rng.choice(N, size=mb, replace=False)
            d = world[b][1][vis] - pose[b][0:2]
            cth, sth = np.cos(pose[b][2]), np.sin(pose[b][2])
            xr = cth * d[:, 0] + sth * d[:, 1] + rng.normal(0, 0.01, mb)
            yr = -sth * d[:, 0] + cth * d[:, 1] + rng.normal(0, 0.01, mb)
            m[k, b] = mb
            idx[k, b, :mb] = vis
            zr[k, b, :mb] = np.sqrt(xr ** 2 + yr ** 2)
            zb[k, b, :mb] = np.arctan2(yr, xr)
    return [w[2] for w in world], lin, ang, idx, zr, zb, m


def check_against_per_step_and_oracle(sd, n, B, starts, means, lin, ang, idx, zr, zb, m, fused, oracle_for=None):
    """Every trajectory against the per-step kernels; the trajectories `oracle_for` (default: all) against the oracle too."""
    plain, _ = run_stream(sd, n, B, starts, means, lin, ang, idx, zr, zb, m, options=[("active_bound", 0), ("fused_cadence", 0)])
    cfg = orc.EkfConfig()
    step = orc.ekf_step_dense if n < 1000 else orc.ekf_step_structured    # (the O(n^2) form where dense takes seconds per step)
    for b in range(B):
        assert orc.rel_fro(fused[b][0], plain[b][0]) < PATH_TOL and orc.rel_fro(fused[b][1], plain[b][1]) < PATH_TOL
        assert np.array_equal(fused[b][1], fused[b][1].T)
        if oracle_for is not None and b not in oracle_for:
            continue
        om, oP = means[b].copy(), starts[b].copy()
        for k in range(len(lin)):
            mb = m[k, b]
            om, oP = step(om, oP, lin[k, b], ang[k, b], idx[k, b, :mb], zr[k, b, :mb], zb[k, b, :mb], cfg)
        assert orc.rel_fro(fused[b][0], om) < TIGHT and orc.rel_fro(fused[b][1], oP) < TIGHT


@pytest.mark.parametrize("N,B,steps,hi", [(150, 4, 60, 8), (150, 3, 40, 16), (90, 1, 70, 3), (600, 30, 20, 8), (1400, 24, 14, 8)])
def test_packed_cadences_with_wandering_landmark_counts(sd, N, B, steps, hi):
    """What the reference's loop produces (src/replay_no_ros.py:280-301, :436: whatever tags the window saw): per trajectory
    and step m ~ uniform{0..hi} landmarks at scattered indices.  Every trajectory walks its own packed sequence -- steps cut
    by a pass, steps that see nothing riding along, trajectories that finish a cadence early idling in the last one -- and
    the bank needs as many passes as its busiest trajectory at 40 landmark updates per pass.  Against the per-step path
    and the oracle; all steps fused.  (N = 600 x 30 and N = 1400 x 24: the one-wave and the throughput shape of the panel
    launch; there the oracle checks two trajectories, the per-step path all.)"""
    n = 3 + 2 * N
    means, lin, ang, idx, zr, zb, m = wandering_stream(N, B, steps, lambda k, b, rng: rng.integers(0, hi + 1), 4200 + hi)
    starts = [dense_start(n, 4300 + t) for t in range(B)]
    fused, (nc, ns) = run_stream(sd, n, B, starts, means, lin, ang, idx, zr, zb, m, options=[("active_bound", 0)])
    need = [cadences_needed(m[:, b]) for b in range(B)]
    assert nc == max(need) and ns == steps
    assert max(need) <= -(-int(m.sum(axis=0).max()) // 40) + 1     # what the busiest trajectory needs at 40 per pass (+ the tail)
    check_against_per_step_and_oracle(sd, n, B, starts, means, lin, ang, idx, zr, zb, m, fused,
                                      oracle_for=None if B <= 4 else (0, B - 1))


@pytest.mark.parametrize("case", range(int(os.environ.get("EKF_FUZZ_CASES", "10"))))   # (a soak: EKF_FUZZ_CASES=200)
def test_random_streams_orders_pieces_and_options_against_the_oracle(sd, case):
    """Everything at once, drawn from a seeded generator: a bank of 1 .. 4 trajectories of 20 .. 320 landmarks, 0 .. `hi`
    landmarks per step and trajectory at scattered indices, dense starting covariances; the run's cadences in one of the three
    orders (chained / look-ahead / plain) or on the per-step kernels; `stream_run` in one call or in random pieces, with or
    without `run_end_flush`, a `flush()` or a download between pieces; the panel launch's shape forced or by size, the row-slab pass
    forced (with W formed from V); the active bound on or off.  Every trajectory against the oracle (the reference-shaped dense step) -- 1e-9 -- and no flag raised."""
    rng = np.random.default_rng(52000 + case)
    N, B = int(rng.integers(20, 321)), int(rng.integers(1, 5))
    steps, hi = int(rng.integers(8, 61)), int(rng.choice([1, 3, 8, 16]))
    n = 3 + 2 * N
    means, lin, ang, idx, zr, zb, m = wandering_stream(N, B, steps, lambda k, b, r: r.integers(0, min(hi, N) + 1), 53000 + case)
    starts = [dense_start(n, 54000 + 10 * case + t) for t in range(B)]
    order = ("chain", "lookahead", "plain", "per_step")[int(rng.integers(0, 4))]
    opts = [("small_state", 0), ("active_bound", int(rng.integers(0, 2)))]
    opts += list(MODES[order]) if order != "per_step" else [("fused_cadence", 0)]
    if rng.integers(0, 2):
        opts.append(("panel_shape", int(rng.integers(1, 4))))
    if rng.integers(0, 5) < 2:                            # the row-slab pass at any size: behind a replay shape of the panel launch
        opts.append(("pass_kernel", 2))                   # it forms its W fragments from V ("w_from_v")
        opts.append(("panel_shape", int(rng.integers(2, 4))))
    if rng.integers(0, 2):
        opts.append(("run_end_flush", 1))
    pieces = [steps]
    if rng.integers(0, 2):
        cuts = sorted(set(int(c) for c in rng.integers(1, steps, size=int(rng.integers(1, 6)))))
        pieces = [b_ - a_ for a_, b_ in zip([0] + cuts, cuts + [steps])]
    with sd.EkfSlam(n, batch=B) as f:
        for name, value in opts:
            f.set_option(name, value)
        for b in range(B):
            f.set_state(means[b], starts[b], b)
        f.stream_upload(lin, ang, idx, zr, zb, m)
        k = 0
        for count in pieces:
            f.stream_run(k, count)
            k += count
            what = int(rng.integers(0, 4))
            if what == 0:
                f.flush()
            elif what == 1:
                assert np.isfinite(f.mean(int(rng.integers(0, B)))).all()
        out = [f.state(b) for b in range(B)]
        assert [f.flags(b) for b in range(B)] == [0] * B, (order, opts, pieces)
        if os.environ.get("EKF_FUZZ_VERBOSE"):
            print("FUZZ", case, order, dict(opts), pieces, "cadences", cadences(sd, f), "w_from_v", sd.load_library().ekf_debug_w_from_v(f._h))
    cfg = orc.EkfConfig()
    for b in range(B):
        om, oP = means[b].copy(), starts[b].copy()
        for k in range(steps):
            mb = m[k, b]
            om, oP = orc.ekf_step_dense(om, oP, lin[k, b], ang[k, b], idx[k, b, :mb], zr[k, b, :mb], zb[k, b, :mb], cfg)
        assert orc.rel_fro(out[b][0], om) < TIGHT and orc.rel_fro(out[b][1], oP) < TIGHT, (order, opts, pieces, b)


@pytest.mark.parametrize("N,B,steps", [(300, 3, 30), (1400, 24, 12), (2100, 2, 12)])
def test_column_gather_beside_the_solve_is_bit_identical(sd, N, B, steps):
    """The mirrored column entries of the panel launch (P(C_u[a], i) for i < C_u[a]: one 16-byte pair per row) fetched by extra
    workgroups of the SOLVE launch and laid down as rows (`col_gather` = 1, the default) against the panel launch gathering
    everything itself (`col_gather` = 0): the same values by another road -- bit for bit, scattered landmarks, the latency and
    the throughput shape of the panel launch, a state of two column panels (N = 2100: pairs next to the panel boundary)."""
    n = 3 + 2 * N
    means, lin, ang, idx, zr, zb, m = wandering_stream(N, B, steps, lambda k, b, rng: rng.integers(0, 9), 6100 + N)
    if N > 2048:
        idx[:, :, 0] = np.where(m > 0, 2046, idx[:, :, 0])     # landmark 2046: columns 4095, 4096 straddle the panels
        for k in range(steps):
            for b in range(B):
                dup = np.nonzero(idx[k, b, 1:m[k, b]] == 2046)[0]
                idx[k, b, 1 + dup] = 2045 - dup                  # (keep the indices of a step distinct)
    starts = [dense_start(n, 6200 + t) for t in range(B)]
    res = {}
    for cg in (1, 0):
        res[cg], (nc, ns) = run_stream(sd, n, B, starts, means, lin, ang, idx, zr, zb, m,
                                       options=[("active_bound", 0), ("lookahead", 0), ("col_gather", cg)])
        assert ns == steps and nc >= 1
    for b in range(B):
        assert np.array_equal(res[1][b][0], res[0][b][0]) and np.array_equal(res[1][b][1], res[0][b][1]), b
    check_against_per_step_and_oracle(sd, n, B, starts, means, lin, ang, idx, zr, zb, m, res[1], oracle_for=(0,))


def test_a_run_longer_than_one_planning_piece(sd):
    """ekf_stream_run plans its packed cadences in pieces of 8192 steps (32 B per cadence and trajectory stay bounded): a run
    of 8300 steps crosses one piece boundary -- every trajectory is brought to the boundary, what is pending there is flushed,
    the next piece starts fused.  Against the per-step path and the oracle."""
    N, B, steps = 40, 2, 8300
    n = 3 + 2 * N
    means, lin, ang, idx, zr, zb, m = wandering_stream(N, B, steps, lambda k, b, rng: rng.integers(0, 3), 6400)
    starts = [dense_start(n, 6500 + t) for t in range(B)]
    opts = [("active_bound", 0), ("small_state", 0)]
    fused, (nc, ns) = run_stream(sd, n, B, starts, means, lin, ang, idx, zr, zb, m, options=opts)
    need = max(cadences_needed(m[:8192, b]) for b in range(B)) + max(cadences_needed(m[8192:, b]) for b in range(B))
    assert nc == need and ns == steps
    plain, _ = run_stream(sd, n, B, starts, means, lin, ang, idx, zr, zb, m, options=opts + [("fused_cadence", 0)])
    cfg = orc.EkfConfig()
    om, oP = means[0].copy(), starts[0].copy()
    for k in range(steps):
        mb = m[k, 0]
        om, oP = orc.ekf_step_dense(om, oP, lin[k, 0], ang[k, 0], idx[k, 0, :mb], zr[k, 0, :mb], zb[k, 0, :mb], cfg)
    for b in range(B):
        assert orc.rel_fro(fused[b][0], plain[b][0]) < 1e-9 and orc.rel_fro(fused[b][1], plain[b][1]) < 1e-9   # (8300 steps of rounding)
    assert orc.rel_fro(fused[0][0], om) < 1e-8 and orc.rel_fro(fused[0][1], oP) < 1e-8


def test_long_runs_without_observations_inside_a_fused_run(sd):
    """Windows in which no tag is seen (the reference's loop then only predicts, src/replay_no_ros.py:435): 45 such steps at
    the head of the stream are one cadence of 40 predictions that appends no rank anywhere in the bank -- no pass follows,
    the panel launch puts the motion noise on the pose diagonal itself -- then observations resume, stop again for 50
    steps in one trajectory only, and the stream ends on steps that see nothing."""
    N, B, steps = 80, 2, 130

    def counts(k, b, rng):
        if k < 45 or k >= 122:
            return 0
        if b == 1 and 60 <= k < 110:
            return 0
        return rng.integers(1, 6)
    n = 3 + 2 * N
    means, lin, ang, idx, zr, zb, m = wandering_stream(N, B, steps, counts, 4400)
    starts = [dense_start(n, 4500 + t) for t in range(B)]
    fused, (nc, ns) = run_stream(sd, n, B, starts, means, lin, ang, idx, zr, zb, m, options=[("active_bound", 0)])
    assert nc == max(cadences_needed(m[:, b]) for b in range(B)) and ns == steps
    check_against_per_step_and_oracle(sd, n, B, starts, means, lin, ang, idx, zr, zb, m, fused)


def test_golden_stream_through_the_cadence(sd):
    """BASELINE config 1 (N = 20, 500 steps, the reference's own outputs in tests/golden/stream_n20_m8.npz) as ONE
    uploaded stream: 100 fused cadences back to back, final mean and covariance against the reference."""
    g = gu.load("stream_n20_m8")
    steps = len(g["lin"])
    with sd.EkfSlam(len(g["mean0"])) as f:
        f.set_option("small_state", 0)                    # (n = 43 would take the small-state path: this test is about the cadence kernels)
        f.set_state_diag(g["mean0"], g["diag0"])
        f.run_stream(g["lin"], g["ang"], g["idx"], g["zr"], g["zb"])
        mu, P = f.state()
        assert cadences(sd, f) == (steps // 5, steps) and f.flags() == 0
    assert orc.rel_fro(mu, g["out_mean"][-1]) < TIGHT
    assert orc.rel_fro(P, g["out_cov"][-1]) < TIGHT


def test_config4_shape_n2000_x32_fused_against_per_step(sd):
    """The benchmarked shape (N = 2000, m = 8, 32 trajectories: the throughput form of the panel launch, the row-slab
    pass behind every full cadence): two full cadences + one of a single step against the per-step path."""
    N, B, steps, m, K = 2000, 32, 11, 8, 2
    n = 3 + 2 * N
    streams = [orc.synthetic_stream(N, steps, m, 40 + t) for t in range(K)]
    starts = [dense_start(n, 7 + t, rank=8) for t in range(K)]
    pick = [streams[b % K] for b in range(B)]
    args = (stack(pick, 2), stack(pick, 3), stack(pick, 4), stack(pick, 5), stack(pick, 6))
    res = {}
    for fused in (1, 0):
        with sd.EkfSlam(n, batch=B) as f:
            f.set_option("active_bound", 0)
            f.set_option("fused_cadence", fused)
            for b in range(B):
                f.set_state(pick[b][0], starts[b % K], b)
            f.run_stream(*args)
            res[fused] = [f.state(b) for b in (0, 1, 30, 31)]
            assert [f.flags(b) for b in range(B)] == [0] * B
            assert cadences(sd, f) == ((3, 11) if fused else (0, 0))
    for a, b in zip(res[1], res[0]):
        assert orc.rel_fro(a[0], b[0]) < PATH_TOL and orc.rel_fro(a[1], b[1]) < PATH_TOL
    assert np.array_equal(res[1][0][1], res[1][2][1]) and np.array_equal(res[1][1][0], res[1][3][0])   # replicas agree bit for bit


def test_trajectories_of_different_size_in_one_bank(sd):
    """A bank whose trajectories hold different numbers of landmarks (n = 3 + 2 N_b each, one capacity): the cadence
    kernels take every trajectory's own size (its rows beyond are never touched).  Each trajectory against its own
    single-trajectory handle of the same capacity -- bit for bit (same launches shapes apart from the batch) -- and one of
    them against the oracle."""
    sizes, steps, m = [400, 250, 120], 23, 8
    n_max = 3 + 2 * max(sizes)
    streams = [orc.synthetic_stream(N, steps, m, 1700 + t) for t, N in enumerate(sizes)]
    starts = [dense_start(3 + 2 * N, 1800 + t) for t, N in enumerate(sizes)]
    args = [np.stack([s[i] for s in streams], axis=1) for i in (2, 3, 4, 5, 6)]
    with sd.EkfSlam(n_max, batch=len(sizes)) as f:
        for b, s in enumerate(streams):
            f.set_state(s[0], starts[b], b)
        f.run_stream(*args)
        together = [f.state(b) for b in range(len(sizes))]
        assert [f.flags(b) for b in range(len(sizes))] == [0] * len(sizes)
        assert cadences(sd, f)[0] >= 4
        assert [f.size(b) for b in range(len(sizes))] == [3 + 2 * N for N in sizes]
    for b, s in enumerate(streams):
        with sd.EkfSlam(n_max, batch=1) as f:
            f.set_state(s[0], starts[b], 0)
            f.run_stream(*[a[:, b:b + 1] for a in args])
            mu, P = f.state(0)
        assert mu.shape == (3 + 2 * sizes[b],)
        assert np.array_equal(mu, together[b][0]) and np.array_equal(P, together[b][1]), b
    cfg = orc.EkfConfig()
    s = streams[2]
    om, oP = s[0].copy(), starts[2].copy()
    for k in range(steps):
        om, oP = orc.ekf_step_dense(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], cfg)
    assert orc.rel_fro(together[2][0], om) < TIGHT and orc.rel_fro(together[2][1], oP) < TIGHT


def test_q_zero_inside_a_cadence_sets_the_flag_and_stays_in_its_trajectory(sd):
    """A landmark exactly at the robot (q = 0, src/replay_no_ros.py:446, :466-469) at the head of a fused cadence (everything after it in the cadence replays NaN): NaN
    like NumPy's 0/0, the sticky EKF_FLAG_NONFINITE on that trajectory -- and the other trajectory of the batch, replayed
    by the same launches, is untouched by it (against the oracle)."""
    N, steps, m = 40, 12, 4
    n = 3 + 2 * N
    cfg = dict(disable_motion_model=True)              # the pose mean stays where the update leaves it: (0, 0) at step 0
    streams = [orc.synthetic_stream(N, steps, m, 1900 + t) for t in range(2)]
    means = [s[0].copy() for s in streams]
    means[0][0:3] = 0.0
    means[0][3 + 2 * int(streams[0][4][0, 0])] = 0.0     # the first landmark observed sits exactly at the robot
    means[0][4 + 2 * int(streams[0][4][0, 0])] = 0.0
    means[1][0:3] = 0.0
    starts = [dense_start(n, 1950 + t) for t in range(2)]
    args = (stack(streams, 2), stack(streams, 3), stack(streams, 4), stack(streams, 5), stack(streams, 6))
    with sd.EkfSlam(n, batch=2, config=sd.EkfConfig(**cfg)) as f:
        f.set_option("active_bound", 0)
        for b in range(2):
            f.set_state(means[b], starts[b], b)
        f.run_stream(*args)
        assert cadences(sd, f)[0] >= 1
        assert f.flags(0) & 1 and f.flags(1) == 0         # EKF_FLAG_NONFINITE
        assert not np.isfinite(f.mean(0)).all()
        mu1, P1 = f.state(1)
    ocfg = orc.EkfConfig(**cfg)
    s = streams[1]
    om, oP = means[1].copy(), starts[1].copy()
    for k in range(steps):
        om, oP = orc.ekf_step_dense(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], ocfg)
    assert orc.rel_fro(mu1, om) < TIGHT and orc.rel_fro(P1, oP) < TIGHT
