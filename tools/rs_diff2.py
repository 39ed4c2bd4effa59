#!/usr/bin/env python3
"""Which partial sum does a wrong entry of pass_kernel=2 hold?  One 80-rank pass after 5 steps; the oracle's covariance
after every step tells how many steps' worth of ranks a wrong entry is missing.  (development aid)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slam_duckietown_amd as sd
from oracle import ekf_oracle as orc

N, B, m, steps = 531, 3, 5, 5
n = 3 + 2 * N
streams = [orc.synthetic_stream(N, steps, m, 30 + t) for t in range(B)]
starts = []
for t in range(B):
    rng = np.random.default_rng(90 + t)
    A = rng.normal(size=(n, 6)) * 0.3
    starts.append(A @ A.T + np.diag(rng.uniform(0.5, 2.0, n)))
cfg = orc.EkfConfig()
hist = []
for b, st in enumerate(streams):
    om, oP = st[0].copy(), starts[b].copy()
    h = [oP.copy()]
    for k in range(steps):
        om, oP = orc.ekf_step_dense(om, oP, st[2][k], st[3][k], st[4][k], st[5][k], st[6][k], cfg)
        h.append(oP.copy())
    hist.append(h)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    with sd.EkfSlam(n, batch=B) as f:
        f.set_option("pass_kernel", 2)
        f.set_option("rank_limit", 80)
        f.set_option("pass_streaming", 1)
        f.set_option("active_bound", 0)
        for b, s in enumerate(streams):
            f.set_state(s[0], starts[b], b)
        for k in range(steps):
            f.step([s[2][k] for s in streams], [s[3][k] for s in streams], [s[4][k] for s in streams],
                   [s[5][k] for s in streams], [s[6][k] for s in streams])
        got = [f.state(b)[1] for b in range(B)]
    for b in range(B):
        d = got[b] - hist[b][steps]
        bad = np.argwhere(np.abs(d) > 1e-11)
        bad = bad[bad[:, 0] <= bad[:, 1]]
        if len(bad):
            print(f"rep {rep} traj {b}: {len(bad)} wrong upper entries, rows {sorted(set(bad[:,0]))[:4]} cols {sorted(set(bad[:,1]))[:8]}")
            for r, c in bad[:4]:
                miss = [hist[b][j][r, c] - hist[b][steps][r, c] for j in range(steps)]
                print(f"   ({r},{c}) wrong-right = {d[r, c]: .3e};  (after step j) - (after step 5), j=0..4: " + " ".join(f"{x: .3e}" for x in miss))
print("done")
