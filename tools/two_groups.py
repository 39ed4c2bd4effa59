#!/usr/bin/env python3
"""Experiment: two handles of B/2 trajectories with their covariance passes half a cadence apart, against one handle of B.
  python3 tools/two_groups.py [--workgroups W]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slam_duckietown_amd as sd
import slam_duckietown_amd.synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--workgroups", type=int, default=0)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--lead", type=int, default=2)
args = ap.parse_args()
N, B, m = 2000, 32, 8
n = 3 + 2 * N
T = args.steps + 20

def make(b0, nb):
    streams = [syn.synthetic_stream(N, T, m, t) for t in range(b0, b0 + nb)]
    f = sd.EkfSlam(n, batch=nb)
    f.set_option("active_bound", 0)
    if args.workgroups:
        f.set_option("pass_workgroups", args.workgroups)
    for b, s in enumerate(streams):
        f.set_state_diag(s[0], s[1], b)
    f.stream_upload(*[np.stack([s[i] for s in streams], 1) for i in (2, 3, 4, 5, 6)])
    return f

one = make(0, B)
one.stream_run(0, 10); one.sync()
t0 = time.perf_counter(); one.stream_run(10, args.steps); one.sync(); t1 = time.perf_counter()
print(f"one handle of {B}: {B * args.steps / (t1 - t0):10.0f} steps/s")
one.close()
a, b = make(0, B // 2), make(B // 2, B // 2)
a.stream_run(0, 10); b.stream_run(0, 10 + args.lead); b.flush(); a.sync(); b.sync()
t0 = time.perf_counter()
pa, pb = 10, 10 + args.lead
end = 10 + args.steps
while pa < end or pb < end:
    if pa < end:
        c = min(5, end - pa); a.stream_run(pa, c); pa += c
    if pb < end:
        c = min(5, end - pb); b.stream_run(pb, c); pb += c
a.sync(); b.sync()
t1 = time.perf_counter()
print(f"two handles of {B // 2}, passes {args.lead} steps apart, pass workgroups {args.workgroups or 'all'}: "
      f"{(B // 2) * (2 * args.steps - args.lead) / (t1 - t0):10.0f} steps/s")
