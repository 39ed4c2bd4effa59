#!/bin/bash
# the column-strip pass (k_flush) at mid sizes: rows per workgroup;  usage: bash tools/rows_sweep.sh "N B" ...
for cfg in "$@"; do
  set -- $cfg
  line="N=$1 B=$2:"
  for r in 0 64 80 96 112 128 160 192 256 336 512; do
    o="--option pass_kernel=0"
    [ $r -gt 0 ] && o="$o --option pass_rows_per_block=$r"
    v=$(python3 bench.py --landmarks $1 --trajectories $2 --no-cpu-baseline --no-single $o 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(f\"{d['value']/1e3:.0f}k/{1e3*(d['roofline'].get('avg_launch_ms') or 0):.0f}us\")")
    line="$line  R=$r $v"
  done
  echo "$line"
done
