#!/usr/bin/env python3
"""Phase timeline of k_solve_cad from a -DCAD_STAMPS build (make -C slam-duckietown_amd/csrc variant_cad TAG=stamps
EXTRA=-DCAD_STAMPS):  EKFSLAM_HIP_VARIANT=stamps python3 tools/cad_stamps.py [--landmarks N] [--trajectories B]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = {0: "w0 start", 1: "w0 (H P) published", 2: "w0 S, K published", 3: "w0 records issued / at b1", 4: "w0 past b1",
         5: "w0 down-date done / at b2", 6: "w0 past b2", 7: "w0 next H read", 8: "w1 start", 9: "w1 at b1", 10: "w1 past b1",
         11: "w1 mean + next Jacobian done / at b2", 12: "w1 innovation done", 13: "w1 mean updated", 15: "-"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--landmarks", type=int, default=2000)
    ap.add_argument("--trajectories", type=int, default=32)
    args = ap.parse_args()
    import slam_duckietown_amd as sd
    import slam_duckietown_amd.synthetic as syn
    lib = sd.load_library()
    B, N = args.trajectories, args.landmarks
    streams = [syn.synthetic_stream(N, 30, 8, t) for t in range(B)]
    f = sd.EkfSlam(3 + 2 * N, batch=B)
    f.set_option("active_bound", 0)
    for b, s in enumerate(streams):
        f.set_state_diag(s[0], s[1], b)
    f.stream_upload(*[np.stack([s[i] for s in streams], 1) for i in (2, 3, 4, 5, 6)])
    f.stream_run(0, 30)
    f.sync()
    size = lib.ekf_debug_cad(f._h, 0, None, 0)
    buf = np.zeros(size // 8, dtype=np.uint64)
    lib.ekf_debug_cad(f._h, 0, buf.ctypes.data_as(C.c_void_p), size)
    # CadHead: 4 + 40 + 84 ints, g[40][2] doubles, then prow
    off = (4 + 40 + 84) * 4 // 8 + 80
    st = buf[off:off + 48].astype(np.int64).reshape(3, 16)
    t0 = st[0, 0]
    names = NAMES
    for s in range(3):
        print(f"-- slot {s} (relative to slot 0's start, cycles)")
        for k in np.argsort(st[s]):
            if int(k) in names and abs(int(st[s, k]) - int(t0)) < 10 ** 7:
                print(f"   {st[s, k] - t0:8d}  {names[int(k)]}")
    f.close()


if __name__ == "__main__":
    main()
