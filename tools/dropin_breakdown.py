#!/usr/bin/env python3
"""Where a drop-in EKF_pose_estimation call spends its time at small N (wall clock around each sub-call)."""
import os, sys, time
from types import SimpleNamespace
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slam_duckietown_amd as sd
import slam_duckietown_amd.synthetic as syn
from slam_duckietown_amd.frontend import associate

N, m, calls = int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 3, 300
mean0, diag0, lin, ang, idx, zr, zb = syn.synthetic_stream(N, calls, m, 0)
ti = {1000 + i: i for i in range(N)}
f = sd.EkfSlam(3 + 2 * N)
f.set_state(mean0, np.diag(diag0))
T = {k: [] for k in ("associate", "step", "size", "state", "flags", "sums")}
mean = mean0
for k in range(calls):
    xr, yr = zr[k] * np.cos(zb[k]), zr[k] * np.sin(zb[k])
    tags = [SimpleNamespace(tag_id=1000 + int(i), pose_R=None, pose_t=np.array([[-y], [0.0], [x]]), pose_err=0.0) for i, x, y in zip(idx[k], xr, yr)]
    t0 = time.perf_counter(); tp = associate([(0.0, tags)], ti, mean, 1.5, ()); t1 = time.perf_counter()
    ks = list(tp.keys())
    f.step(lin[k], ang[k], ks, [tp[q][4] for q in ks], [tp[q][5] for q in ks]); t2 = time.perf_counter()
    n = f.size(); t3 = time.perf_counter()
    mean, cov = f.state(); t4 = time.perf_counter()
    fl = f.flags(); t5 = time.perf_counter()
    s = np.concatenate([cov.sum(axis=0), cov.sum(axis=1)]); t6 = time.perf_counter()
    for name, a, b in (("associate", t0, t1), ("step", t1, t2), ("size", t2, t3), ("state", t3, t4), ("flags", t4, t5), ("sums", t5, t6)):
        T[name].append(b - a)
print(f"N={N} m={m}: " + "  ".join(f"{k} {np.median(v[20:]) * 1e6:6.1f} us" for k, v in T.items()) +
      f"   total {sum(np.median(v[20:]) for v in T.values()) * 1e6:6.1f} us")
