#!/usr/bin/env python3
"""Where a drop-in EKF_pose_estimation call spends its time: the real function with wall clocks around its parts (host
association, upload, the fused step + download), medians over the calls.
  python3 tools/dropin_breakdown.py [N m [N m ...]] [--after LEG]
Several (N, m) pairs run one after the other in ONE process (like bench.py's drop_in leg); --after LEG runs a secondary leg
of bench.py in this process first (leg-order effects)."""
import os, sys, time
from types import SimpleNamespace
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slam_duckietown_amd as sd
import slam_duckietown_amd.synthetic as syn
from slam_duckietown_amd import ekf_bindings as eb

if "--after" in sys.argv:
    import argparse, bench
    i = sys.argv.index("--after")
    bench.secondary_leg(sys.argv[i + 1], argparse.Namespace(gpus=1, steps=200, warmup=20, landmarks=2000, obs=8, trajectories=32,
                                                            option=[], leg=None, no_cpu_baseline=True, no_single=False))
    del sys.argv[i:i + 2]
T = {}


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        T.setdefault(name, []).append(time.perf_counter() - t0)
        return r
    return w


eb.associate = timed("associate", eb.associate)
sd.EkfSlam.set_state = timed("set_state", sd.EkfSlam.set_state)
sd.EkfSlam.step_state = timed("step_state", sd.EkfSlam.step_state)


def run(N, m):
    T.clear()
    calls = 300 if N <= 100 else 60
    mean0, diag0, lin, ang, idx, zr, zb = syn.synthetic_stream(N, calls, m, 0)
    ti = {1000 + i: i for i in range(N)}
    mean, cov = mean0.copy(), np.diag(diag0)
    total = []
    for k in range(calls):
        xr, yr = zr[k] * np.cos(zb[k]), zr[k] * np.sin(zb[k])
        tags = [SimpleNamespace(tag_id=1000 + int(i), pose_R=None, pose_t=np.array([[-y], [0.0], [x]]), pose_err=0.0)
                for i, x, y in zip(idx[k], xr, yr)]
        t0 = time.perf_counter()
        mean, cov, _ = sd.EKF_pose_estimation(ang[k], lin[k], mean, cov, 0.7, [(0.0, tags)], ti)
        total.append(time.perf_counter() - t0)
    med = {k: np.median(v[len(v) // 6:]) * 1e6 * (len(v) / calls) for k, v in T.items()}
    tot = np.median(total[10:]) * 1e6
    print(f"N={N} m={m} fetch_spin={os.environ.get('EKFSLAM_HIP_FETCH_SPIN', 'default')}: total {tot:6.1f} us = " +
          "  ".join(f"{k} {v:5.1f}" for k, v in med.items()) + f"  rest {tot - sum(med.values()):5.1f}")


pairs = [int(a) for a in sys.argv[1:]] or [12, 3]
for q in range(0, len(pairs), 2):
    run(pairs[q], pairs[q + 1] if q + 1 < len(pairs) else 3)
