#!/bin/bash
# round-4 scratch: the pass without its V loads (timing only)
mkdir -p gpurun_out/r4a
O=gpurun_out/r4a/ft8.txt
: > $O
for V in "" skipv; do
  export EKFSLAM_HIP_VARIANT=$V
  [ -z "$V" ] && unset EKFSLAM_HIP_VARIANT
  for fe in 1 4 5; do
    python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 --option flush_every=$fe >> $O 2>&1 || exit 1
  done
  python3 -W ignore tools/flush_time.py --landmarks 2000 --trajectories 32 >> $O 2>&1 || exit 1
  python3 -W ignore tools/flush_time.py --landmarks 3000 --trajectories 16 >> $O 2>&1 || exit 1
done
cat $O
