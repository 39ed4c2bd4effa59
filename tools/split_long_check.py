#!/usr/bin/env python3
"""Long runs of the single-launch step (k_step_split, k_panels_split) against the two-launch path: mean and covariance
must be bit-identical after every configuration (a stale read through the hand-over would show here).
  python3 tools/split_long_check.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam_duckietown_amd as sd
from oracle import ekf_oracle as orc
for (N, B, m, steps) in ((1200, 16, 8, 120), (2000, 12, 8, 60), (300, 3, 5, 200), (2000, 1, 8, 200)):
    streams = [orc.synthetic_stream(N, steps, m, 70 + t) for t in range(B)]
    out = {}
    for fused in (0, 1):
        with sd.EkfSlam(3 + 2 * N, batch=B) as f:
            f.set_option("fused_step", fused)
            for b, s in enumerate(streams):
                f.set_state_diag(s[0], s[1], b)
            f.run_stream(*[np.stack([s[i] for s in streams], 1) for i in (2, 3, 4, 5, 6)])
            out[fused] = [f.state(b) for b in range(B)]
            assert [f.flags(b) for b in range(B)] == [0] * B
    same = all(np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]) for a, c in zip(out[0], out[1]))
    print(N, B, m, steps, "bit-identical" if same else "MISMATCH")
    assert same
