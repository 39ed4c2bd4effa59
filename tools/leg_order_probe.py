#!/usr/bin/env python3
"""Why bench.py's config-5 leg reads ~8 % slower behind the headline leg of the same process (round 4): allocation placement or
device state?  One mode per process:
  first            the N = 8000 leg first thing in a fresh process
  after_headline   ... after the headline leg (32 x N = 2000: 4.4 GB allocated and freed)
  twice_after      ... and then once more
  prealloc         the N = 8000 handle is CREATED first (fresh allocation), the headline leg runs and frees its handle, then
                   the N = 8000 handle is timed
  hold_dummy       after the headline leg a 6 GB dummy allocation is made and held, then the N = 8000 leg"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import slam_duckietown_amd as sd
import slam_duckietown_amd.synthetic as syn
import slam_duckietown_amd.sharding as shard
grp = shard.RankGroup()
mode = sys.argv[1]


def headline():
    bench.time_filter(sd, syn, shard, grp, 0, list(range(32)), 2000, 8, 200, 20, options=["active_bound=0"])


def leg(tag):
    out = bench.config5_leg(sd, syn, shard, grp, 0, 8)
    print(mode, tag, "config5 dense pass us", round(out["dense"]["pass_avg_launch_ms"] * 1e3, 1), "steps/s", round(out["dense"]["value"]))


def timed_handle(f, streams_args, warmup=20, steps=100):
    f.stream_run(0, warmup)
    f.flush()
    f.sync()
    f.profile_enable(True)
    f.stream_run(warmup, steps)
    f.flush()
    ms, cnt = f.profile_read()
    return ms / cnt * 1e3


if mode == "first":
    leg("")
elif mode == "after_headline":
    headline()
    leg("")
elif mode == "twice_after":
    headline()
    leg("1st")
    leg("2nd")
elif mode == "hold_dummy":
    headline()
    hip = C.CDLL("libamdhip64.so")
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(6 << 30)) == 0
    leg("(6 GB held)")
elif mode in ("after_headline_nolook", "first_nolook"):
    if mode == "after_headline_nolook":
        headline()
    N = 8000
    s = syn.synthetic_stream(N, 120, 8, 0)
    f = sd.EkfSlam(3 + 2 * N)
    f.set_option("active_bound", 0)
    f.set_option("lookahead", 0)
    f.set_state_diag(s[0], s[1])
    f.stream_upload(*[np.stack([s[i]], 1) for i in (2, 3, 4, 5, 6)])
    f.sync()
    print(mode, "pass us", round(timed_handle(f, None), 1))
elif mode == "prealloc":
    N = 8000
    s = syn.synthetic_stream(N, 120, 8, 0)
    f = sd.EkfSlam(3 + 2 * N)
    f.set_option("active_bound", 0)
    f.set_state_diag(s[0], s[1])
    f.stream_upload(*[np.stack([s[i]], 1) for i in (2, 3, 4, 5, 6)])
    f.sync()
    headline()
    print(mode, "pass us", round(timed_handle(f, None), 1))
