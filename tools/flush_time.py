#!/usr/bin/env python3
"""Average duration of the covariance-pass kernel (HIP events on the handle's stream) for a library variant.
  EKFSLAM_HIP_VARIANT=<tag> python3 tools/flush_time.py [--landmarks N] [--trajectories B] [--obs m] [--option k=v ...]
Diagnostic variants (make -C slam-duckietown_amd/csrc variant TAG=... EXTRA=-D...) may compute wrong covariances:
nothing is checked here, only timed."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--landmarks", type=int, default=2000)
    ap.add_argument("--trajectories", type=int, default=32)
    ap.add_argument("--obs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--option", action="append", default=[])
    ap.add_argument("--nmax", type=int, default=0, help="capacity of the handle (sets the row stride ld = ceil(nmax / 64) * 64)")
    ap.add_argument("--dense-start", action="store_true", help="start from a dense covariance (diagonal + rank 8): every entry of "
                    "V / W is non-zero from the first step on -- fp64 MFMA power, and with it the clock, depends on the operands")
    args = ap.parse_args()
    import slam_duckietown_amd as sd
    import slam_duckietown_amd.synthetic as syn
    n = 3 + 2 * args.landmarks
    B = args.trajectories
    streams = [syn.synthetic_stream(args.landmarks, args.steps + 10, args.obs, t) for t in range(B)]
    f = sd.EkfSlam(max(n, args.nmax | 1), batch=B)
    f.set_option("active_bound", 0)
    for o in args.option:
        k, v = o.split("=")
        f.set_option(k, int(v))
    for b, s in enumerate(streams):
        if args.dense_start:
            rng = np.random.default_rng(b)
            A = rng.normal(size=(n, 8)) * 0.3
            P = A @ A.T
            P[np.arange(n), np.arange(n)] += rng.uniform(0.5, 2.0, n)
            f.set_state(s[0], P, b)
            del P
        else:
            f.set_state_diag(s[0], s[1], b)
    f.stream_upload(*[np.stack([s[i] for s in streams], 1) for i in (2, 3, 4, 5, 6)])
    f.stream_run(0, 10)
    f.flush()
    f.sync()
    f.profile_enable(True)
    f.stream_run(10, args.steps)
    f.flush()
    ms, cnt = f.profile_read()
    tri = n * (n + 1) / 2.0
    per = ms / max(cnt, 1)
    print(f"variant={os.environ.get('EKFSLAM_HIP_VARIANT', 'default'):10s} options={args.option} nmax={args.nmax} {'dense start ' if args.dense_start else ''}launches={cnt} "
          f"avg={per * 1e3:8.1f} us  {B * 16.0 * tri / (per * 1e-3) / 1e12:6.3f} TB/s algorithmic")
    f.close()


if __name__ == "__main__":
    main()
