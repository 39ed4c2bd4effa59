# kernel-trace of bench.py for B=32 and B=1 (fused cadence): per-kernel averages
export TMPDIR=/tmp
for B in 32 1; do
  rm -rf gpurun_out/r3kt_$B; mkdir -p gpurun_out/r3kt_$B
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3kt_$B -o run -- python3 bench.py --no-cpu-baseline --no-single --trajectories $B "$@" > gpurun_out/r3kt_$B/log.txt 2>&1
  echo "== B=$B $@"; python3 tools/kernel_times.py gpurun_out/r3kt_$B | grep "ekf::"
done
