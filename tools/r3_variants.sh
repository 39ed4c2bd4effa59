# kernel times of the cadence kernels for library variants: bash tools/r3_variants.sh v1 v2 ...   ("" = product)
export TMPDIR=/tmp
for v in "$@"; do
  [ "$v" = "default" ] && v=""
  rm -rf gpurun_out/kv_$v; mkdir -p gpurun_out/kv_$v
  EKFSLAM_HIP_VARIANT=$v rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kv_$v -o run -- python3 bench.py --no-cpu-baseline --no-single --steps 100 > gpurun_out/kv_$v/log.txt 2>&1
  echo "== variant '$v'"; python3 tools/kernel_times.py gpurun_out/kv_$v | grep "solve_cad\|panels_cad"
done
