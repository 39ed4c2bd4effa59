#!/usr/bin/env python3
"""Host-driven EkfSlam.step per call (bench.py's online_step leg) over map and bank sizes, beside the uploaded-stream rate."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import slam_duckietown_amd as sd, slam_duckietown_amd.synthetic as syn, slam_duckietown_amd.sharding as shard
grp = shard.RankGroup()
print("    N     B    online steps/s   us/call  host us/call    stream steps/s")
for N, B in [(12, 1), (12, 64), (38, 256), (100, 1), (100, 32), (100, 256), (500, 1), (500, 32), (500, 128), (1000, 1), (1000, 32),
             (2000, 1), (2000, 8), (2000, 32), (4000, 1), (4000, 8)]:
    o = bench.online_step_leg(sd, syn, 0, N, B, 8, 100, 10, [])
    dt, _, _, _ = bench.time_filter(sd, syn, shard, grp, 0, list(range(B)), N, 8, 100, 20, profile_leg=False)
    print(f"{N:6d} {B:5d} {o['value']:14.0f} {o['ms_per_call'] * 1e3:9.1f} {o['host_enqueue_ms_per_call'] * 1e3:10.1f} {B * 100 / dt:16.0f}")
