#!/usr/bin/env python3
"""Download of a whole state into a pinned array: k_pack_dense (a kernel mirrors and writes over PCIe) against the mirror
pass + the runtime's rectangle copy (SDMA), for several sizes; optionally behind a bench leg in the same process
(the runtime's copy is slower by half for some sizes once a process has used more streams).
  python3 tools/download_paths.py [leg]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import slam_duckietown_amd as sd, slam_duckietown_amd.synthetic as syn
leg = sys.argv[1] if len(sys.argv) > 1 else None
if leg:
    bench.secondary_leg(leg, argparse.Namespace(gpus=1, steps=200, warmup=20, landmarks=2000, obs=8, trajectories=32,
                                                option=[], leg=None, no_cpu_baseline=True, no_single=False))
print(f"behind {leg or 'nothing'}:   N   MB    k_pack_dense ms (GB/s)    mirror + rectangle copy ms (GB/s)")
for N in (100, 200, 350, 500, 700, 1000, 1500, 2000, 3000):
    n = 3 + 2 * N
    s = syn.synthetic_stream(N, 4, 8, 0)
    f = sd.EkfSlam(n)
    f.set_state_diag(s[0], s[1])
    f.step(s[2][0], s[3][0], s[4][0], s[5][0], s[6][0])
    res = []
    for pack in (2, 0):
        f.set_option("pack_dense", pack)
        t = []
        for _ in range(12):
            t0 = time.perf_counter()
            mu, P = f.state()
            t.append(time.perf_counter() - t0)
            del mu, P
        res.append(np.median(t[3:]))
    mb = 8 * n * n / 1e6
    print(f"            {N:5d} {mb:6.1f}    {res[0] * 1e3:8.3f} ({mb / res[0] / 1e3:5.1f})          {res[1] * 1e3:8.3f} ({mb / res[1] / 1e3:5.1f})")
    f.close()
