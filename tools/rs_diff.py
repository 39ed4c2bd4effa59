#!/usr/bin/env python3
"""Where do pass_kernel=0 and pass_kernel=2 differ?  (development aid)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slam_duckietown_amd as sd
from oracle import ekf_oracle as orc

N, B, m, steps = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (531, 3, 5, 11)))
limit, streaming = (int(x) for x in (sys.argv[5:7] if len(sys.argv) > 6 else (80, 1)))
n = 3 + 2 * N
streams = [orc.synthetic_stream(N, steps, m, 30 + t) for t in range(B)]
starts = []
for t in range(B):
    rng = np.random.default_rng(90 + t)
    A = rng.normal(size=(n, 6)) * 0.3
    starts.append(A @ A.T + np.diag(rng.uniform(0.5, 2.0, n)))
cfg = orc.EkfConfig()
oracle = []
for b, st in enumerate(streams):
    om, oP = st[0].copy(), starts[b].copy()
    for k in range(steps):
        om, oP = orc.ekf_step_dense(om, oP, st[2][k], st[3][k], st[4][k], st[5][k], st[6][k], cfg)
    oracle.append(oP)
out = {}
for rep in range(4):
    for kernel in (0, 2):
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
            out[kernel] = [f.state(b) for b in range(B)]
    for b in range(B):
        d = out[0][b][1] - out[2][b][1]
        bad = np.argwhere(d != 0)
        print(f"rep {rep} traj {b}: differing entries {len(bad)}  max |diff| {np.abs(d).max():.3e}  mean diff {np.abs(out[0][b][0]-out[2][b][0]).max():.3e}")
        if len(bad):
            rows, cols = bad[:, 0], bad[:, 1]
            up = bad[rows <= cols]
            print("   upper-triangle entries:", len(up), " rows", sorted(set(up[:, 0] // 16 * 16))[:12], " cols", sorted(set(up[:, 1] // 64 * 64))[:12])
            print("   first few:", [(int(r), int(c), float(d[r, c])) for r, c in up[:6]])
            e0 = max(abs(out[0][b][1][r, c] - oracle[b][r, c]) for r, c in up)
            e2 = max(abs(out[2][b][1][r, c] - oracle[b][r, c]) for r, c in up)
            print(f"   at those entries: |kernel0 - oracle| <= {e0:.3e}   |kernel2 - oracle| <= {e2:.3e}")
