#!/bin/bash
# round-4: the covariance pass on DENSE operands (every entry of V / W non-zero) against the benchmark's block-diagonal start
mkdir -p gpurun_out/r4j
O=gpurun_out/r4j/dense.txt
: > $O
for fe in 1 2 3 4 5; do
  python3 -W ignore tools/flush_time.py --landmarks 2000 --trajectories 32 --option flush_every=$fe >> $O 2>&1 || exit 1
  python3 -W ignore tools/flush_time.py --landmarks 2000 --trajectories 32 --option flush_every=$fe --dense-start >> $O 2>&1 || exit 1
done
EKFSLAM_HIP_VARIANT=nopmem python3 -W ignore tools/flush_time.py --landmarks 2000 --trajectories 32 >> $O 2>&1
EKFSLAM_HIP_VARIANT=nopmem python3 -W ignore tools/flush_time.py --landmarks 2000 --trajectories 32 --dense-start >> $O 2>&1
for w in 20 260 520; do
  python3 bench.py --no-cpu-baseline --no-single --warmup $w 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench warmup $w: value', round(d['value']), 'pass us', round(d['roofline']['avg_launch_ms']*1e3,1), 'frac', round(d['roofline']['frac'],3))" >> $O
done
cat $O
