#!/usr/bin/env python3
"""Step / flush / download of a single N = 500 filter, alone and behind a bench leg in the same process."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import slam_duckietown_amd as sd, slam_duckietown_amd.synthetic as syn
N = 500
early = None
if "--handle-first" in sys.argv:               # the filter is created BEFORE the leg runs: its memory is placed early
    sys.argv.remove("--handle-first")
    early = sd.EkfSlam(3 + 2 * N)
if len(sys.argv) > 1:
    bench.secondary_leg(sys.argv[1], argparse.Namespace(gpus=1, steps=200, warmup=20, landmarks=2000, obs=8, trajectories=32,
                                                        option=[], leg=None, no_cpu_baseline=True, no_single=False))
def where():
    import ctypes
    cpu = ctypes.CDLL(None).sched_getcpu()
    node = "?"
    try:
        for d in os.listdir("/sys/devices/system/node"):
            if d.startswith("node") and os.path.exists(f"/sys/devices/system/node/{d}/cpu{cpu}"):
                node = d
    except OSError:
        pass
    return f"cpu {cpu} ({node}), affinity {len(os.sched_getaffinity(0))} cpus"
print("now on", where(), file=sys.stderr)
if os.environ.get("PROBE_PIN"):
    lo, hi = os.environ["PROBE_PIN"].split("-")
    os.sched_setaffinity(0, set(range(int(lo), int(hi) + 1)))
    print("pinned:", where(), file=sys.stderr)
s = syn.synthetic_stream(N, 120, 8, 0)
f = early or sd.EkfSlam(3 + 2 * N)
f.set_state_diag(s[0], s[1])
T = {"step": [], "flush": [], "state": []}
f.profile_enable(True)
for k in range(100):
    t0 = time.perf_counter(); f.step(s[2][k], s[3][k], s[4][k], s[5][k], s[6][k]); f.sync(); t1 = time.perf_counter()
    f.flush(); f.sync(); t2 = time.perf_counter()
    mu, P = f.state(); t3 = time.perf_counter()
    T["step"].append(t1 - t0); T["flush"].append(t2 - t1); T["state"].append(t3 - t2)
import ctypes as C
dp = C.POINTER(C.c_double)
n = 3 + 2 * N
mu2, P2 = np.empty(n), np.empty((n, n))
P2[:] = 0.0
lib = sd.load_library()
tp = []
for _ in range(30):
    t0 = time.perf_counter(); lib.ekf_download_state(f._h, 0, mu2.ctypes.data_as(dp), P2.ctypes.data_as(dp), n); tp.append(time.perf_counter() - t0)
T["state into a reused pageable array"] = tp
tb = []
for _ in range(30):
    t0 = time.perf_counter(); blk = f.covariance_block(0, 0, n, n); tb.append(time.perf_counter() - t0)
T["download_block (no mirror)"] = tb
ms, cnt = f.profile_read()
print(f"behind {sys.argv[1] if len(sys.argv) > 1 else 'nothing':8s}{' (handle created first)' if early else ''}: " + "  ".join(f"{k} {np.median(v[len(v)//5:]) * 1e6:6.1f} us" for k, v in T.items()),
      f"  pass kernel {ms / max(cnt, 1) * 1e3:.1f} us x {cnt}  {f.last_pass()}")
