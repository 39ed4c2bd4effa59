#!/bin/bash
# EKF steps/s against the map size N and the bank size B (bench.py's headline leg only: uploaded streams, m = 8, block-diagonal
# start, 20 warm-up + 200 timed steps; trajectory-steps per second over the bank; the pass kernel and its fraction of the HBM peak).
echo "    N     B     steps/s    ms/step   pass kernel                              avg us   frac of 8 TB/s"
for nb in "12 1" "12 256" "12 2048" "20 1" "20 256" "38 1" "38 256" "45 256" "45 1024" "64 256" "64 1024" "100 1" "100 32" "100 256" "500 1" "500 32" "500 128" "1000 1" "1000 32" "1000 64" "2000 1" "2000 8" "2000 16" "2000 32" "2000 48" "4000 1" "4000 8" "8000 1" "8000 2" "10000 1"; do
  set -- $nb
  python3 bench.py --landmarks $1 --trajectories $2 --no-cpu-baseline --no-single 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
tail = f\"{str(r.get('kernel'))[:40]:40s} {1e3*(r.get('avg_launch_ms') or 0):7.1f}   {r['frac']:.3f}\" if r.get('avg_launch_ms') else '(small-state path: no pass)'
print(f\"{$1:6d} {$2:5d} {d['value']:11.0f} {d['ms_per_step']:10.5f}   \" + tail)
"
done
