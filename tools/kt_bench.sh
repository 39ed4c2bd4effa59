# kernel-trace averages of bench.py's headline (B = 32) and single trajectory (B = 1):  bash tools/kt_bench.sh <tag> [bench options]
export TMPDIR=/tmp
tag=$1; shift
OUT=gpurun_out/kt_$tag
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/b32 -o run -- python3 bench.py --no-cpu-baseline --no-single "$@" > $OUT/b32.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/b1 -o run -- python3 bench.py --no-cpu-baseline --no-single --trajectories 1 "$@" > $OUT/b1.log 2>&1
python3 tools/kernel_times.py $OUT/b32 $OUT/b1 | grep -v "copyBuffer\|fill_diag" > $OUT/kernel_times.txt
tail -1 $OUT/b32.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('b32 steps/s', round(d['value']), 'pass ms', round(d['roofline']['avg_launch_ms'],4))" >> $OUT/kernel_times.txt
tail -1 $OUT/b1.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('b1 steps/s', round(d['value']))" >> $OUT/kernel_times.txt
cat $OUT/kernel_times.txt
