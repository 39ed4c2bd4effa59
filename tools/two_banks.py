#!/usr/bin/env python3
"""Experiment: N=2000, 32 trajectories as ONE bank of 32 against TWO banks of 16 (own streams, enqueued from one
host thread): does the second bank's solve / panel launches fill the gaps of the first bank's pass?"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slam_duckietown_amd as sd
import slam_duckietown_amd.synthetic as syn

N, m, steps, warm = 2000, 8, 200, 20
n = 3 + 2 * N
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
opts = [o.split("=") for o in sys.argv[3:]]
streams = [syn.synthetic_stream(N, steps + warm, m, t) for t in range(B)]
banks = []
per = B // G
for g in range(G):
    f = sd.EkfSlam(n, batch=per)
    f.set_option("active_bound", 0)
    for k, v in opts:
        f.set_option(k, int(v))
    mine = streams[g * per:(g + 1) * per]
    for b, s in enumerate(mine):
        f.set_state_diag(s[0], s[1], b)
    f.stream_upload(*[np.stack([s[i] for s in mine], 1) for i in (2, 3, 4, 5, 6)])
    f.stream_run(0, warm)
    f.flush()
    banks.append(f)
for f in banks:
    f.sync()
t0 = time.perf_counter()
from slam_duckietown_amd.sharding import run_banks
run_banks(banks, warm, steps)                 # slices of 50 steps, bank after bank: both queues are fed from the start
for f in banks:
    f.flush()
for f in banks:
    f.sync()
dt = time.perf_counter() - t0
print(f"B={B} in {G} bank(s) of {per} {opts}: {B * steps / dt:.0f} steps/s  ({dt * 1e3:.1f} ms)")
for f in banks:
    assert not any(f.flags(b) for b in range(per))
    f.close()
