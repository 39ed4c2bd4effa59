# chained solves against the round-3 look-ahead, single trajectory (BASELINE config 3): steps/s and kernel timelines
#   bash tools/chain_probe.sh <tag> [chain values, default "1 0"]
export TMPDIR=/tmp
tag=${1:-chain}
vals=${2:-"1 0"}
OUT=gpurun_out/$tag
rm -rf $OUT; mkdir -p $OUT
for c in $vals; do
  python3 bench.py --leg single_trajectory --option chain=$c > $OUT/bench_chain$c.json 2>$OUT/bench_chain$c.err || exit 1
  rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_chain$c -o run -- python3 bench.py --leg single_trajectory --option chain=$c > $OUT/kt_chain$c.log 2>&1 || exit 1
  python3 tools/cad_timeline.py $OUT/kt_chain$c 2 > $OUT/timeline_chain$c.txt
  cat $OUT/bench_chain$c.json $OUT/timeline_chain$c.txt
done
