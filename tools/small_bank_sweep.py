#!/usr/bin/env python3
"""Small-state path, banks of N = 20 filters (one workgroup each; 3 resident per CU at 148 VGPRs): steps/s against the bank size.
The 64 distinct synthetic trajectories are repeated to fill the bank."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slam_duckietown_amd as sd, slam_duckietown_amd.synthetic as syn

N, m, steps = 20, 8, 500
base = [syn.synthetic_stream(N, steps, m, t) for t in range(64)]
for B in [int(a) for a in sys.argv[1:]] or [256, 512, 768, 1536, 3072, 6144]:
    f = sd.EkfSlam(3 + 2 * N, batch=B)
    for b in range(B):
        f.set_state_diag(base[b % 64][0], base[b % 64][1], b)
    cols = [np.ascontiguousarray(np.stack([base[b % 64][i] for b in range(B)], 1)) for i in (2, 3, 4, 5, 6)]
    f.stream_upload(*cols)
    f.stream_run(0, 20)
    f.sync()
    t0 = time.perf_counter()
    f.stream_run(0, steps)
    f.sync()
    dt = time.perf_counter() - t0
    assert not any(f.flags(b) for b in (0, B - 1))
    print(f"bank of {B:5d}: {B * steps / dt / 1e6:7.2f} M steps/s  ({dt * 1e3:.2f} ms for {steps} steps each)")
    f.close()
