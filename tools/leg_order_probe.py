#!/usr/bin/env python3
"""Does a secondary leg of bench.py read differently behind other legs of the same process?
  python3 tools/leg_order_probe.py drop_in                 (alone)
  python3 tools/leg_order_probe.py config5 online_step drop_in   (in this order, one process; the LAST one is printed in full)"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

legs = sys.argv[1:]
args = argparse.Namespace(gpus=1, steps=200, warmup=20, landmarks=2000, obs=8, trajectories=32, option=[], leg=None,
                          no_cpu_baseline=True, no_single=False)
out = {}
for name in legs:
    out = bench.secondary_leg(name, args)
print(json.dumps(out))
