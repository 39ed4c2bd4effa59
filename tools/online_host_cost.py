#!/usr/bin/env python3
"""Host cost of one online `EkfSlam.step` call, without back-pressure from the device: bursts of 12
calls behind a sync (the input ring has 16 slots: a longer burst waits for the device and measures ITS time, not the host's).
Per batch size: the whole Python call, the bare C call (`ekf_step` through ctypes with pointers made once), their difference
= the binding's staging.   python3 tools/online_host_cost.py [--landmarks N]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--landmarks", type=int, default=2000)
    args = ap.parse_args()
    import slam_duckietown_amd as sd
    import slam_duckietown_amd.synthetic as syn
    from slam_duckietown_amd import ekf_bindings as eb
    lib = sd.load_library()
    N, m, burst, reps = args.landmarks, 8, 12, 12
    for B in (1, 32):
        streams = [syn.synthetic_stream(N, burst * reps * 2 + 20, m, t) for t in range(B)]
        f = sd.EkfSlam(3 + 2 * N, batch=B)
        f.set_option("active_bound", 0)
        for b, s in enumerate(streams):
            f.set_state_diag(s[0], s[1], b)
        cols = [[np.ascontiguousarray(np.stack([s[i][k] for s in streams])) for k in range(len(streams[0][2]))] for i in (2, 3, 4, 5, 6)]

        def py_call(k):
            if B == 1:
                f.step(cols[0][k][0], cols[1][k][0], cols[2][k][0], cols[3][k][0], cols[4][k][0])
            else:
                f.step(cols[0][k], cols[1][k], cols[2][k], cols[3][k], cols[4][k])

        mm = np.full(B, m, dtype=np.int32)

        def c_call(k):
            rc = lib.ekf_step(f._h, eb._p(cols[0][k]), eb._p(cols[1][k]), eb._p(cols[2][k], eb._ip), eb._p(cols[3][k]),
                              eb._p(cols[4][k]), eb._p(mm, eb._ip), m)
            assert rc == 0

        ptrs = None
        out = {}
        for name, fn in (("python", py_call), ("c_abi_incl_pointer_making", c_call)):
            for k in range(10):
                fn(k)
            f.flush()
            f.sync()
            ts, k = [], 10
            for _ in range(reps):
                t0 = time.perf_counter()
                for _ in range(burst):
                    fn(k)
                    k += 1
                ts.append((time.perf_counter() - t0) / burst)
                f.flush()
                f.sync()
            out[name] = float(np.median(ts)) * 1e6
        # the bare C call with every pointer made beforehand
        k0 = 10 + burst * reps
        pp = [(eb._p(cols[0][k]), eb._p(cols[1][k]), eb._p(cols[2][k], eb._ip), eb._p(cols[3][k]), eb._p(cols[4][k]))
              for k in range(k0, k0 + burst)]
        pm = eb._p(mm, eb._ip)
        f.sync()
        t0 = time.perf_counter()
        for a in pp:
            lib.ekf_step(f._h, a[0], a[1], a[2], a[3], a[4], pm, m)
        out["c_abi_bare"] = (time.perf_counter() - t0) / burst * 1e6
        f.flush()
        f.sync()
        f.close()
        print(f"N={N} x {B}: us per call  " + "  ".join(f"{k} {v:.1f}" for k, v in out.items()))


if __name__ == "__main__":
    main()
