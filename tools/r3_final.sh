# full GPU suite, then the committed kernel-trace summaries and the default bench line
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/final/pytest_gpu.txt 2>&1 || { tail -20 gpurun_out/final/pytest_gpu.txt; exit 1; }
tail -3 gpurun_out/final/pytest_gpu.txt
OUT=gpurun_out/final
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b32 -o run -- python3 bench.py --no-cpu-baseline --no-single > $OUT/stats_b32.log 2>&1
echo "stats b32 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b1 -o run -- python3 bench.py --no-cpu-baseline --no-single --trajectories 1 > $OUT/stats_b1.log 2>&1
echo "stats b1 done"
python3 tools/kernel_times.py $OUT/stats_b32 $OUT/stats_b1 > $OUT/kernel_times.txt
cat $OUT/kernel_times.txt
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 -c "import json; d=json.load(open('$OUT/bench_default.json')); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['single_trajectory']['value'], d['obs_1_per_step']['value'], d['config5']['dense']['value'], d['config5']['skip_unobserved']['value'], d['dense_propagate']['TFLOPs_fp64'], d['cpu_baseline']['value'])"
