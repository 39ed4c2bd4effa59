#!/bin/bash
# round-4: what copying wanted columns out of the pass's tiles costs the pass: RS_COLBUF = 1 / 2 slots per tile
mkdir -p gpurun_out/r4g
O=gpurun_out/r4g/colbuf.txt
: > $O
for rep in 1 2 3; do
  for V in current colbuf1 colbuf2; do
    if [ "$V" = current ]; then unset EKFSLAM_HIP_VARIANT; else export EKFSLAM_HIP_VARIANT=$V; fi
    python3 tools/pass_drift.py --fused 1 >> $O 2>&1 || exit 1
  done
done
cat $O
