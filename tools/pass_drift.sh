#!/bin/bash
# round-4: the pass (N = 2000 x 32, 80 ranks) for round-2 head / round-3 head / current (builtin store with the guard tied to
# its data registers, early loads of tile t+2) / current with the late loads, behind per-step kernels and fused cadence, ONE box
mkdir -p gpurun_out/r4d
O=gpurun_out/r4d/drift3.txt
: > $O
for rep in 1 2 3 4; do
  for V in r2head r3head current laten3; do
    if [ "$V" = current ]; then unset EKFSLAM_HIP_VARIANT; else export EKFSLAM_HIP_VARIANT=$V; fi
    python3 tools/pass_drift.py --fused 0 >> $O 2>&1 || exit 1
  done
  for V in r3head current laten3; do
    if [ "$V" = current ]; then unset EKFSLAM_HIP_VARIANT; else export EKFSLAM_HIP_VARIANT=$V; fi
    python3 tools/pass_drift.py --fused 1 >> $O 2>&1 || exit 1
  done
done
unset EKFSLAM_HIP_VARIANT
python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 >> $O 2>&1
EKFSLAM_HIP_VARIANT=laten3 python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 >> $O 2>&1
python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 >> $O 2>&1
EKFSLAM_HIP_VARIANT=laten3 python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 >> $O 2>&1
cat $O
