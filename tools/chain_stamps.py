#!/usr/bin/env python3
"""Phase timeline of k_chain_cad from a -DCHAIN_STAMPS build (make -C slam-duckietown_amd/csrc variant_cad TAG=chstamps
EXTRA=-DCHAIN_STAMPS):  EKFSLAM_HIP_VARIANT=chstamps python3 tools/chain_stamps.py [--landmarks N]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = ["start", "records + positions staged, C zeroed", "coefficients", "Linv; gathered rows there", "loads issued, Linv C",
         "A X in LDS", "Linv A X", "E solved", "-F + mean partials", "F^T E", "block written"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--landmarks", type=int, default=2000)
    args = ap.parse_args()
    import slam_duckietown_amd as sd
    import slam_duckietown_amd.synthetic as syn
    lib = sd.load_library()
    N = args.landmarks
    s = syn.synthetic_stream(N, 60, 8, 0)
    f = sd.EkfSlam(3 + 2 * N, batch=1)
    f.set_option("active_bound", 0)
    f.set_state_diag(s[0], s[1], 0)
    f.stream_upload(*[np.stack([s[i]], 1) for i in (2, 3, 4, 5, 6)])
    f.stream_run(0, 60)
    f.sync()
    buf = np.zeros(128 + 32, dtype=np.float64)
    lib.ekf_debug_snapshot(f._h, 0, 5, buf.ctypes.data_as(C.POINTER(C.c_double)), buf.size)
    st = buf[128:].view(np.uint64).astype(np.int64)
    for k, name in enumerate(NAMES):
        print(f"  {st[k] - st[0]:8d} cycles  ~ {(st[k] - st[0]) / 2400.0:7.2f} us at 2.4 GHz  {name}")
    f.close()


if __name__ == "__main__":
    main()
