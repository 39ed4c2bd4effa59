#!/bin/bash
# N = 2000 x 1 (BASELINE config 3): the covariance pass under its launch options
for o in "" "--option lookahead=0" "--option pass_kernel=2" "--option pass_kernel=2 --option lookahead=0" "--option pass_streaming=1" "--option pass_streaming=0"; do
  python3 tools/flush_time.py --trajectories 1 --steps 100 $o
done
for r in 48 64 80 96 112 128 160 192 256; do
  python3 tools/flush_time.py --trajectories 1 --steps 100 --option pass_rows_per_block=$r --option lookahead=0
done
