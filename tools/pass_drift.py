#!/usr/bin/env python3
"""Where the covariance pass's 13 - 30 us go: the same pass (N = 2000 x 32, 80 ranks) timed on ONE box
  * behind the fused cadence's panel launch and behind the per-step kernels (fused_cadence = 1 / 0),
  * for the round-2 head, the round-3 head and the current library (built from `git archive` into
    libekfslam_hip_r2head.so / _r3head.so, selected with EKFSLAM_HIP_VARIANT),
with the shader clock sampled while the passes run.  One (variant, mode) per process:
  EKFSLAM_HIP_VARIANT=r3head python3 tools/pass_drift.py --fused 1
Prints one line."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--landmarks", type=int, default=2000)
    ap.add_argument("--trajectories", type=int, default=32)
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    import warnings
    warnings.simplefilter("ignore")
    import slam_duckietown_amd as sd
    import slam_duckietown_amd.synthetic as syn
    import bench
    n = 3 + 2 * args.landmarks
    B = args.trajectories
    streams = [syn.synthetic_stream(args.landmarks, args.steps + 20, 8, t) for t in range(B)]
    f = sd.EkfSlam(n, batch=B)
    f.set_option("active_bound", 0)
    note = ""
    try:
        f.set_option("fused_cadence", args.fused)
    except sd.EkfError:
        note = "(no fused cadence in this library: per-step kernels)"
    for b, s in enumerate(streams):
        f.set_state_diag(s[0], s[1], b)
    f.stream_upload(*[np.stack([s[i] for s in streams], 1) for i in (2, 3, 4, 5, 6)])
    f.stream_run(0, 20)
    f.flush()
    f.sync()
    clk = bench.ClockSampler(0, period_s=0.001)
    f.profile_enable(True)
    clk.start()
    f.timer_begin()
    f.stream_run(20, args.steps)
    f.flush()
    total_ms = f.timer_end()
    clk.stop()
    ms, cnt = f.profile_read()
    c = clk.summary() or {"min": 0, "mean": 0, "max": 0, "samples": 0}
    tri = n * (n + 1) / 2.0
    per = ms / max(cnt, 1)
    print(f"variant={os.environ.get('EKFSLAM_HIP_VARIANT', 'current'):8s} fused={args.fused} pass avg {per * 1e3:7.1f} us over {cnt} launches "
          f"({B * 16.0 * tri / (per * 1e-3) / 1e12:5.3f} TB/s)  {B * args.steps / (total_ms * 1e-3) / 1e3:6.1f} k steps/s  "
          f"sclk min/mean/max {c['min']:.0f}/{c['mean']:.0f}/{c['max']:.0f} MHz ({c['samples']} samples) {f.last_pass()} {note}")
    f.close()


if __name__ == "__main__":
    main()
