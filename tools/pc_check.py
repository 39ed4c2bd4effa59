#!/usr/bin/env python3
"""k_flush_pc against k_flush: same stream through both forms of the covariance pass."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slam_duckietown_amd as sd
from slam_duckietown_amd import synthetic as syn

def run(N, B, steps, m, kernel, rank_limit=80):
    n = 3 + 2 * N
    f = sd.EkfSlam(n, batch=B)
    f.set_option("pass_kernel", kernel)
    f.set_option("rank_limit", rank_limit)
    f.set_option("active_bound", 0)
    streams = [syn.synthetic_stream(N, steps, m, t) for t in range(B)]
    for b, s in enumerate(streams):
        f.set_state_diag(s[0], s[1], b)
    lin = np.stack([s[2] for s in streams], 1); ang = np.stack([s[3] for s in streams], 1)
    idx = np.stack([s[4] for s in streams], 1); zr = np.stack([s[5] for s in streams], 1); zb = np.stack([s[6] for s in streams], 1)
    f.stream_upload(lin, ang, idx, zr, zb)
    t0 = time.perf_counter()
    f.stream_run(0, steps); f.flush(); f.sync()
    dt = time.perf_counter() - t0
    out = [f.state(b) for b in (0, B - 1)]
    flags = [f.flags(b) for b in range(B)]
    f.close()
    return out, flags, dt

for (N, B, steps, m, rl) in ((20, 1, 23, 8, 80), (100, 3, 17, 8, 64), (700, 2, 12, 8, 80), (2000, 2, 11, 8, 80), (2000, 2, 7, 8, 32)):
    a, fa, _ = run(N, B, steps, m, 0, rl)
    c, fc, _ = run(N, B, steps, m, 1, rl)
    ok = all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(a, c))
    err = max(np.abs(x[1] - y[1]).max() for x, y in zip(a, c))
    print(f"N={N} B={B} steps={steps} rank_limit={rl}: identical={ok} max|dP|={err:.3e} flags classic={fa} pc={fc}", flush=True)
