#!/bin/bash
# round-4: (real data) loads of tile t+2 issued earlier (RS_EARLY_N3), and the share order, against the default
mkdir -p gpurun_out/r4a
O=gpurun_out/r4a/ft10.txt
: > $O
for V in "" earlyn3 "" earlyn3; do
  export EKFSLAM_HIP_VARIANT=$V
  [ -z "$V" ] && unset EKFSLAM_HIP_VARIANT
  python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 >> $O 2>&1 || exit 1
  python3 -W ignore tools/flush_time.py --landmarks 2000 --trajectories 32 >> $O 2>&1 || exit 1
  python3 -W ignore tools/flush_time.py --landmarks 3000 --trajectories 16 >> $O 2>&1 || exit 1
done
unset EKFSLAM_HIP_VARIANT
for so in 1 0 1 0; do
  python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 --option pass_share_order=$so >> $O 2>&1 || exit 1
  python3 -W ignore tools/flush_time.py --landmarks 4000 --trajectories 4 --option pass_share_order=$so >> $O 2>&1 || exit 1
done
cat $O
