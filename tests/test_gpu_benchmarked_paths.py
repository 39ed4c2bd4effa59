"""GPU parity of the EXACT kernel instantiations and sizes bench.py runs (BASELINE configs 3, 4, 5), with the path
asserted (a regression in the cadence / pass planning that silently fell back to other kernels must turn these red):

* N=2000 x 32 trajectories, default options as bench.py sets them: two full fused cadences of 5 steps (`k_solve_cad` +
  `k_panels_cad<8,4>`), each followed by the streaming 80-rank row-slab pass `k_flush_rs<20,true>`, plus a one-step
  tail; dense covariances, against `oracle.ekf_step_structured`;
* the streaming 80-rank column-strip instantiation `k_flush<15,5,true>` forced at N=300 against the reference-shaped
  dense path;
* `ekf_predict_dense` at n=4003 against NumPy dgemm;
* N=8000 x 1 as bench.py's `config5.dense` leg runs it (row-slab pass on equal static shares on 255 workgroups, the next
  cadence's solve beside it) against the oracle on the active part; with and without the active bound, bit for bit;
* two handles driven from two host threads.
Reference: src/replay_no_ros.py:430 (propagation), :473-480 (gain, mean and covariance update).
"""
import ctypes as C
import threading

import numpy as np
import pytest

from oracle import ekf_oracle as orc

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


def debug_counters(sd, f):
    """(fused cadences, steps they covered, look-aheads, pieces of the longest static share of the last pass or 0)."""
    lib = sd.load_library()
    a, b = C.c_long(), C.c_long()
    assert lib.ekf_debug_cadences(f._h, C.byref(a), C.byref(b)) == 0
    return a.value, b.value, lib.ekf_debug_lookaheads(f._h), lib.ekf_debug_last_pass_shares(f._h)


def dense_start(n, seed):
    """A dense SPD covariance (diagonal + rank 8) so that every tile of the pass receives a non-zero update."""
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(n, 8)) * 0.3
    P = A @ A.T
    P[np.arange(n), np.arange(n)] += rng.uniform(0.5, 2.0, n)
    return P


def test_config4_shard_n2000_x32_default_cadence(sd):
    """BASELINE config 4's per-GPU shard exactly as bench.py runs it (N=2000, m=8, 32 trajectories, active bound
    off, default cadence = 5 steps per 80-rank streaming pass): 11 steps = two full fused cadences of 5 steps, each with its
    `k_flush_rs<20,true>` launch, plus a one-step tail -- asserted, not assumed.  Trajectories 0-2 start from three
    different dense covariances and are compared with the O(n^2) oracle; trajectory b > 2 repeats trajectory b % 3, so
    all 32 are checked -- bit for bit -- against an oracle-checked one, wherever they sit in the launch."""
    N, steps, m, B, K = 2000, 11, 8, 32, 3
    n = 3 + 2 * N
    cfg = orc.EkfConfig()
    streams = [orc.synthetic_stream(N, steps, m, 40 + t) for t in range(K)]
    starts = [dense_start(n, 7 + t) for t in range(K)]
    ref = []
    for t in range(K):
        om, oP = streams[t][0].copy(), starts[t].copy()
        for k in range(steps):
            s = streams[t]
            om, oP = orc.ekf_step_structured(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], cfg)
        ref.append((om, oP))
    pick = [streams[b % K] for b in range(B)]
    with sd.EkfSlam(n, batch=B) as f:
        f.set_option("active_bound", 0)
        for b in range(B):
            f.set_state(pick[b][0], starts[b % K], b)
        f.run_stream(np.stack([s[2] for s in pick], 1), np.stack([s[3] for s in pick], 1),
                     np.stack([s[4] for s in pick], 1), np.stack([s[5] for s in pick], 1),
                     np.stack([s[6] for s in pick], 1))
        # the path bench.py times: two full fused cadences covering 10 steps, the last full pass the row-slab kernel at 20
        # k-tiles (the third cadence -- the one-step tail -- is still pending here: its pass runs with the first download below)
        cad, covered, _, shares = debug_counters(sd, f)
        assert (cad, covered) == (3, 11)
        assert f.last_pass() == "ekf::k_flush_rs<20, true, false, true>" and shares == 0
        got = {}
        for b in range(B):
            mu, P = f.state(b)
            assert f.flags(b) == 0
            assert np.array_equal(P, P.T)
            d = np.diag(P)
            assert (d > 0).all() and (d[3:] <= np.diag(starts[b % K])[3:] * (1 + 1e-12)).all()   # updates never add variance
            if b < K:
                close(mu, ref[b][0])
                close(P, ref[b][1])
                close(P.sum(axis=1), ref[b][1].sum(axis=1))
                got[b] = (mu, P)
            else:
                assert np.array_equal(mu, got[b % K][0]) and np.array_equal(P, got[b % K][1])


def _oracle_stream(args):
    """(worker process) one trajectory's stream through the O(n^2) oracle from a block-diagonal start."""
    N, steps, m, tid = args
    s = orc.synthetic_stream(N, steps, m, tid)
    cfg = orc.EkfConfig()
    om, oP = s[0].copy(), np.diag(s[1])
    for k in range(steps):
        om, oP = orc.ekf_step_structured(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], cfg)
    return om, oP


def test_steady_state_leg_n2000_x32_as_benchmarked(sd):
    """The regime bench.py's `steady_state` leg (and `value_steady_state` / `roofline.frac_steady_state`) times: the headline
    workload -- N = 2000, m = 8, 32 trajectories, the library's defaults, block-diagonal start -- BEHIND a full sweep of the
    landmarks, where every landmark has been observed, the covariance is dense and V / W carry no exact zeros.  270 steps =
    the sweep (250) + four more cadences, as one uploaded stream; trajectories 0 and 1 (two different streams) against
    `oracle.ekf_step_structured` over all 270 steps (two worker processes: ~0.2 s per step each), trajectory b > 1 -- a
    replica of trajectory b % 2 elsewhere in the launches -- bit for bit against it; the path asserted: 54 fused cadences
    covering every step, the last full pass `k_flush_rs<20, true, false>` on the work queues.
    Reference: src/replay_no_ros.py:368-480 (the step), :476-480 (mean and covariance update)."""
    import concurrent.futures as cf
    import slam_duckietown_amd.synthetic as syn
    N, steps, m, B, K = 2000, 270, 8, 32, 2
    n = 3 + 2 * N
    with cf.ProcessPoolExecutor(max_workers=K) as pool:
        futures = [pool.submit(_oracle_stream, (N, steps, m, t)) for t in range(K)]   # (run beside the GPU part below)
        streams = [syn.synthetic_stream(N, steps, m, t) for t in range(K)]
        pick = [streams[b % K] for b in range(B)]
        with sd.EkfSlam(n, batch=B) as f:
            for b in range(B):
                f.set_state_diag(pick[b][0], pick[b][1], b)
            f.run_stream(*[np.stack([s[i] for s in pick], 1) for i in (2, 3, 4, 5, 6)])
            cad, covered, _, shares = debug_counters(sd, f)
            assert (cad, covered) == (steps * m // 40, steps)                # 54 cadences of 5 steps
            f.flush()
            assert f.last_pass() == "ekf::k_flush_rs<20, true, false, true>" and shares == 0
            got = {}
            for b in (0, 1, 2, 3, 16, 31):
                mu, P = f.state(b)
                assert f.flags(b) == 0 and np.array_equal(P, P.T)
                if b < K:
                    got[b] = (mu, P)
                else:
                    assert np.array_equal(mu, got[b % K][0]) and np.array_equal(P, got[b % K][1])
        # dense behind the sweep: every landmark is correlated with the pose and with its neighbours
        assert np.count_nonzero(got[0][1][3:, 0]) == n - 3
        for t in range(K):
            om, oP = futures[t].result()
            close(got[t][0], om)
            close(got[t][1], oP)
            close(got[t][1].sum(axis=1), oP.sum(axis=1))


def test_variable_m_leg_n2000_x32_as_benchmarked(sd):
    """bench.py's `variable_m` leg at its own size and options (N = 2000, 32 trajectories, m ~ U{0..8} per trajectory and step at
    scattered indices, active bound off): packed cadences with every trajectory on its own cursor, the column gather beside
    the solve, the row-slab pass behind every cadence -- the paths asserted, trajectories 0 and 1 against the O(n^2) oracle,
    and trajectory b > 1 -- a replica of trajectory b % 2 sitting elsewhere in the launches, among neighbours at other steps --
    bit for bit against it."""
    import slam_duckietown_amd.synthetic as syn
    N, steps, B, K = 2000, 14, 32, 2
    n = 3 + 2 * N
    cfg = orc.EkfConfig()
    streams = [syn.variable_stream(N, steps, 0, 8, 70 + t) for t in range(K)]
    starts = [dense_start(n, 17 + t) for t in range(K)]
    ref = []
    for t in range(K):
        s = streams[t]
        om, oP = s[0].copy(), starts[t].copy()
        for k in range(steps):
            mk = s[7][k]
            om, oP = orc.ekf_step_structured(om, oP, s[2][k], s[3][k], s[4][k][:mk], s[5][k][:mk], s[6][k][:mk], cfg)
        ref.append((om, oP))
    pick = [streams[b % K] for b in range(B)]
    with sd.EkfSlam(n, batch=B) as f:
        f.set_option("active_bound", 0)
        for b in range(B):
            f.set_state(pick[b][0], starts[b % K], b)
        f.run_stream(*[np.stack([s[i] for s in pick], 1) for i in (2, 3, 4, 5, 6, 7)])
        cad, covered, _, shares = debug_counters(sd, f)
        need = max(-(-int(s[7].sum()) // 40) for s in streams)
        assert covered == steps and need <= cad <= need + 1          # 40 landmark updates per pass (+ the tail)
        assert f.last_pass().startswith("ekf::k_flush_rs<") and shares == 0
        got = {}
        for b in (0, 1, 2, 3, 30, 31):
            mu, P = f.state(b)
            assert f.flags(b) == 0 and np.array_equal(P, P.T)
            if b < K:
                close(mu, ref[b][0])
                close(P, ref[b][1])
                got[b] = (mu, P)
            else:
                assert np.array_equal(mu, got[b % K][0]) and np.array_equal(P, got[b % K][1])


def test_streaming_80_rank_pass_small(sd):
    """The same instantiation (nontemporal, 15 k-tiles of the V strip in registers + 5 in LDS) forced on a small
    state, every step against the reference-shaped dense path."""
    N, steps, m, B = 300, 12, 8, 2
    cfg = orc.EkfConfig()
    streams = [orc.synthetic_stream(N, steps, m, 60 + t) for t in range(B)]
    n = 3 + 2 * N
    ref = [(s[0].copy(), dense_start(n, 3 + b)) for b, s in enumerate(streams)]
    with sd.EkfSlam(n, batch=B) as f:
        f.set_option("pass_streaming", 1)
        f.set_option("rank_limit", 80)
        for b in range(B):
            f.set_state(ref[b][0], ref[b][1], b)
        for k in range(steps):
            f.step([s[2][k] for s in streams], [s[3][k] for s in streams], [s[4][k] for s in streams],
                   [s[5][k] for s in streams], [s[6][k] for s in streams])
            for b, s in enumerate(streams):
                ref[b] = orc.ekf_step_dense(ref[b][0], ref[b][1], s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], cfg)
            if k in (4, 9, 11):                       # after the first pass, the second, and with a tail pending
                for b in range(B):
                    mu, P = f.state(b)
                    close(mu, ref[b][0])
                    close(P, ref[b][1])


def test_predict_dense_n4003(sd):
    """`ekf_predict_dense` at the size bench.py times it (n = 4003) against NumPy dgemm (src/replay_no_ros.py:430
    with a general F)."""
    n = 4003
    rng = np.random.default_rng(5)
    P0 = dense_start(n, 11)
    F = np.eye(n) + rng.normal(size=(n, n)) * (0.1 / np.sqrt(n))
    Nq = rng.normal(size=(n, 8)) * 0.05
    Q = Nq @ Nq.T + np.diag(rng.uniform(0.01, 0.1, n))
    with sd.EkfSlam(n) as f:
        f.set_state(np.zeros(n), P0)
        f.predict_dense(F, Q)
        P = f.covariance()
    close(P, F @ P0 @ F.T + Q, 1e-12)


def test_config5_n8000_active_bound_bit_identical(sd):
    """BASELINE config 5 at full size (n = 16003): the skip-unobserved pass (active bound on) against the dense pass
    (off) -- bit for bit on the whole matrix -- plus the O(n^2) oracle on sampled rows."""
    N, steps, m = 8000, 12, 8
    mean0, diag0, lin, ang, idx, zr, zb = orc.synthetic_stream(N, steps, m, 9)
    idx = (idx * 37 + 5) % 3000                      # observations stay inside the first 3000 landmarks
    for k in range(steps):
        assert len(set(idx[k].tolist())) == m
    n = len(mean0)
    top = 3 + 2 * (int(idx.max()) + 1)
    out = []
    for bound in (1, 0):
        with sd.EkfSlam(n) as f:
            f.set_option("active_bound", bound)
            # (with the bound on, the pass covers 6003 state indices and is the column-strip kernel, and such a launch
            #  would take the look-ahead -- whose gathered block sums the pending ranks in another order than the pass:
            #  equal to rounding, tests/test_gpu_cadence.py, but this test is about the bound being EXACT)
            f.set_option("lookahead", 0)
            f.set_state_diag(mean0, diag0)
            f.run_stream(lin, ang, idx, zr, zb)
            mu, P = f.state()
            assert f.flags() == 0
            out.append((mu, P))
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
    P = out[0][1]
    del out[1:]
    assert not P[top:, :top].any() and not P[:top, top:].any()            # never correlated: untouched
    assert np.array_equal(np.diag(P)[top:], diag0[top:])
    cfg = orc.EkfConfig()
    om, oP = mean0[:top].copy(), np.diag(diag0[:top])                     # the active part is a closed system
    for k in range(steps):
        om, oP = orc.ekf_step_structured(om, oP, lin[k], ang[k], idx[k], zr[k], zb[k], cfg)
    close(out[0][0][:top], om)
    close(P[:top, :top], oP)


def test_config5_n8000_dense_leg_as_benchmarked(sd):
    """What bench.py's `config5.dense` leg runs, at full size and with ITS options (active_bound = 0, everything else
    default): fused cadences, the row-slab pass on equal static shares on one workgroup per CU minus one, the next
    cadence's solve beside it (look-ahead) -- each asserted -- against the O(n^2) oracle on the active part (the
    observations stay inside the first 3000 landmarks, so the rest of the block-diagonal start must come back untouched)."""
    N, steps, m = 8000, 12, 8
    mean0, diag0, lin, ang, idx, zr, zb = orc.synthetic_stream(N, steps, m, 9)
    idx = (idx * 37 + 5) % 3000
    n = len(mean0)
    top = 3 + 2 * (int(idx.max()) + 1)
    with sd.EkfSlam(n) as f:
        f.set_option("active_bound", 0)
        f.set_state_diag(mean0, diag0)
        f.run_stream(lin, ang, idx, zr, zb)
        cad, covered, lookaheads, shares = debug_counters(sd, f)
        assert (cad, covered) == (3, 12) and lookaheads >= 1
        assert f.last_pass() == "ekf::k_flush_rs<20, true, true, false>" and shares >= 1
        mu, P = f.state()
        assert f.flags() == 0
    assert np.array_equal(P, P.T)
    assert not P[top:, :top].any()                                        # never correlated: exactly zero
    off = P[top:, top:].copy()
    assert np.array_equal(np.diag(off), diag0[top:])
    off[np.arange(n - top), np.arange(n - top)] = 0.0
    assert not off.any()
    cfg = orc.EkfConfig()
    om, oP = mean0[:top].copy(), np.diag(diag0[:top])
    for k in range(steps):
        om, oP = orc.ekf_step_structured(om, oP, lin[k], ang[k], idx[k], zr[k], zb[k], cfg)
    close(mu[:top], om)
    close(P[:top, :top], oP)
    assert np.array_equal(mu[top:], mean0[top:])


def test_two_handles_from_two_host_threads(sd):
    """one handle per host thread, driven concurrently (ctypes releases the GIL).  Each
    handle's result equals its own single-threaded run bit for bit."""
    N, steps, m = 300, 40, 8
    streams = [orc.synthetic_stream(N, steps, m, 70 + t) for t in range(2)]
    n = 3 + 2 * N

    def run(s, out, slot, barrier=None):
        with sd.EkfSlam(n, batch=1) as f:
            f.set_state_diag(s[0], s[1])
            if barrier is not None:
                barrier.wait()
            for k in range(steps):
                f.step(s[2][k], s[3][k], s[4][k], s[5][k], s[6][k])
                if k % 7 == 0:
                    f.mean()                       # a blocking call in the middle of the other thread's work
            out[slot] = f.state() + (f.flags(),)

    alone, together = [None, None], [None, None]
    for t in range(2):
        run(streams[t], alone, t)
    barrier = threading.Barrier(2)
    threads = [threading.Thread(target=run, args=(streams[t], together, t, barrier)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
        assert not th.is_alive()
    for t in range(2):
        assert together[t] is not None and together[t][2] == 0
        assert np.array_equal(alone[t][0], together[t][0]) and np.array_equal(alone[t][1], together[t][1])
    cfg = orc.EkfConfig()
    s = streams[0]
    om, oP = s[0].copy(), np.diag(s[1])
    for k in range(steps):
        om, oP = orc.ekf_step_dense(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], cfg)
    close(together[0][0], om)
    close(together[0][1], oP)


def test_banks_driven_from_one_host_thread(sd):
    """a GPU's trajectories as several handles ("banks", sharding.split_banks), their uploaded
    streams enqueued alternately from ONE host thread (sharding.run_banks) so that one bank's solve and panel launches
    run under another bank's covariance pass.  Every trajectory must come out exactly as when its bank runs alone: the
    banks share nothing but the device."""
    from slam_duckietown_amd.sharding import split_banks, run_banks
    N, steps, m, total = 700, 33, 8, 7
    n = 3 + 2 * N
    streams = [orc.synthetic_stream(N, steps, m, 90 + t) for t in range(total)]
    groups = split_banks(list(range(total)), 4)
    assert [len(g) for g in groups] == [4, 3]

    def make(ids):
        f = sd.EkfSlam(n, batch=len(ids))
        mine = [streams[t] for t in ids]
        for b, s in enumerate(mine):
            f.set_state_diag(s[0], s[1], b)
        f.stream_upload(*[np.stack([s[i] for s in mine], axis=1) for i in (2, 3, 4, 5, 6)])
        return f

    alone = {}
    for ids in groups:
        with make(ids) as f:
            run_banks([f], 0, steps, slice_steps=6)    # (the same slices: where a cadence ends depends on them)
            for b, t in enumerate(ids):
                alone[t] = f.state(b)
    banks = [make(ids) for ids in groups]
    try:
        run_banks(banks, 0, steps, slice_steps=6)
        for f in banks:
            f.sync()
        for f, ids in zip(banks, groups):
            for b, t in enumerate(ids):
                mu, P = f.state(b)
                assert f.flags(b) == 0
                assert np.array_equal(mu, alone[t][0]) and np.array_equal(P, alone[t][1]), t
    finally:
        for f in banks:
            f.close()
    cfg = orc.EkfConfig()
    s = streams[5]
    om, oP = s[0].copy(), np.diag(s[1])
    for k in range(steps):
        om, oP = orc.ekf_step_structured(om, oP, s[2][k], s[3][k], s[4][k], s[5][k], s[6][k], cfg)
    close(alone[5][0], om)
    close(alone[5][1], oP)
