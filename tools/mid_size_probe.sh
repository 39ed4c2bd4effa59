#!/bin/bash
# mid-size banks: the column-strip pass (k_flush) against the row-slab pass (k_flush_rs) -- steps/s of the headline leg and the
# pass's average launch time;  usage: bash tools/mid_size_probe.sh "N B" ...
for cfg in "$@"; do
  set -- $cfg
  line="N=$1 B=$2"
  for o in "--option pass_kernel=0" "--option pass_kernel=2" ""; do
    r=$(python3 bench.py --landmarks $1 --trajectories $2 --no-cpu-baseline --no-single $o 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print(f\"{d['value']:9.0f} steps/s {1e3*(r.get('avg_launch_ms') or 0):7.1f} us {str(r.get('kernel'))[5:30]}\")
")
    line="$line | ${o:-auto}: $r"
  done
  echo "$line"
done
