# kernel-trace averages of one flush_time.py run for a library variant:  bash tools/kt.sh <variant> [flush_time args]
export TMPDIR=/tmp
v=$1; shift
rm -rf gpurun_out/kt_$v; mkdir -p gpurun_out/kt_$v
EKFSLAM_HIP_VARIANT=$v rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_$v -o run -- python3 tools/flush_time.py "$@" > gpurun_out/kt_$v/log.txt 2>&1
echo "== variant '$v' $@"; python3 tools/kernel_times.py gpurun_out/kt_$v | grep "ekf::k_solve\|ekf::k_panels\|ekf::k_step\|k_flush"
