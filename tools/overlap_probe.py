#!/usr/bin/env python3
"""How much do G handles (one HIP stream each) on one GPU overlap?  N=2000, m=8, B trajectories in total."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slam_duckietown_amd as sd
from slam_duckietown_amd import synthetic as syn
import bench

N, m, B, steps, warm = 2000, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 200, 20
for G in (1, 2, 4):
    n = 3 + 2 * N
    hs = []
    for g in range(G):
        ids = list(range(g * B // G, (g + 1) * B // G))
        streams, lin, ang, idx, zr, zb = bench.make_streams(syn, ids, N, steps + warm, m)
        f = sd.EkfSlam(n, batch=len(ids), device=0)
        f.set_option("active_bound", 0)
        for b, s in enumerate(streams):
            f.set_state_diag(s[0], s[1], b)
        f.stream_upload(lin, ang, idx, zr, zb)
        hs.append(f)
    for f in hs:
        f.stream_run(0, warm); f.flush()
    for f in hs:
        f.sync()
    t0 = time.perf_counter()
    # interleave the enqueue in chunks so that every stream has work queued
    for c in range(0, steps, 8):
        for f in hs:
            f.stream_run(warm + c, min(8, steps - c))
    for f in hs:
        f.flush()
    for f in hs:
        f.sync()
    dt = time.perf_counter() - t0
    print(f"G={G}: {B * steps / dt:.0f} steps/s  ({dt / steps * 1e3:.3f} ms per step of {B})", flush=True)
    for f in hs:
        f.close()
