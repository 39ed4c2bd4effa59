# Kernel-trace stats of the observation-shape legs (round 5): constant m = 4 against m ~ uniform{0..8} at scattered indices,
# m = 5, m = 12; plus the headline (B = 32) and the single trajectory (B = 1).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_shapes
rm -rf $OUT; mkdir -p $OUT
for L in constant_m4 variable_m obs_5 obs_12; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$L -o run -- python3 bench.py --leg $L > $OUT/$L.log 2>&1
  echo "$L done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b32 -o run -- python3 bench.py --no-cpu-baseline --no-single > $OUT/stats_b32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b1 -o run -- python3 bench.py --no-cpu-baseline --no-single --trajectories 1 > $OUT/stats_b1.log 2>&1
python3 tools/kernel_times.py $OUT/constant_m4 $OUT/variable_m $OUT/obs_5 $OUT/obs_12 $OUT/stats_b32 $OUT/stats_b1 > $OUT/kernel_times.txt
cat $OUT/kernel_times.txt
