#!/usr/bin/env python3
"""How long the host takes to ENQUEUE a stream against how long the device takes to run it:
   python3 tools/host_enqueue_time.py [trajectories] [landmarks] [name=value ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slam_duckietown_amd as sd
import slam_duckietown_amd.synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
opts = [o.split("=") for o in sys.argv[3:]]
m, steps, warm = 8, 200, 20
n = 3 + 2 * N
streams = [syn.synthetic_stream(N, steps + warm, m, t) for t in range(B)]
f = sd.EkfSlam(n, batch=B)
f.set_option("active_bound", 0)
for k, v in opts:
    f.set_option(k, int(v))
for b, s in enumerate(streams):
    f.set_state_diag(s[0], s[1], b)
f.stream_upload(*[np.stack([s[i] for s in streams], 1) for i in (2, 3, 4, 5, 6)])
f.stream_run(0, warm)
f.flush()
f.sync()
for rep in range(3):
    t0 = time.perf_counter()
    f.stream_run(warm, steps)
    t1 = time.perf_counter()
    f.flush()
    f.sync()
    t2 = time.perf_counter()
    print(f"B={B} N={N} {opts}: enqueue {1e3 * (t1 - t0):.2f} ms, until done {1e3 * (t2 - t0):.2f} ms "
          f"({B * steps / (t2 - t0):.0f} steps/s; {1e6 * (t1 - t0) / (steps / 5):.1f} us of host time per 5-step cadence)")
    # (the stream is replayed from the same state region: only timing matters here)
f.close()
