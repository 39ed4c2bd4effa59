#!/usr/bin/env python3
"""Online step latency / back-to-back throughput of one small filter on the small-state path and on the general path:
  python3 tools/step_latency.py [N ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slam_duckietown_amd as sd, slam_duckietown_amd.synthetic as syn

sizes = [int(a) for a in sys.argv[1:]] or [12, 20, 30, 38, 45, 64]
for N in sizes:
    m = min(8, N)
    s = syn.synthetic_stream(N, 400, m, 0)
    for small in (1, 0):
        f = sd.EkfSlam(3 + 2 * N)
        f.set_option("small_state", small)
        f.set_state_diag(s[0], s[1])
        for k in range(50):
            f.step(s[2][k], s[3][k], s[4][k], s[5][k], s[6][k])
        f.sync()
        t = []
        for k in range(50, 250):
            t0 = time.perf_counter()
            f.step(s[2][k], s[3][k], s[4][k], s[5][k], s[6][k])
            f.sync()
            t.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        for k in range(250, 400):
            f.step(s[2][k], s[3][k], s[4][k], s[5][k], s[6][k])
        f.sync()
        thr = (time.perf_counter() - t0) / 150
        f.stream_upload(*[np.stack([s[i]], 1) for i in (2, 3, 4, 5, 6)])
        f.sync()
        t0 = time.perf_counter()
        f.stream_run(0, 400)
        f.sync()
        st = (time.perf_counter() - t0) / 400
        print(f"N={N:3d} m={m} small_state={small}: step+sync {np.median(t) * 1e6:6.1f} us   back-to-back {thr * 1e6:6.1f} us/step   "
              f"uploaded stream {st * 1e6:6.1f} us/step")
        f.close()
