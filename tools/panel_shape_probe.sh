# kernel-trace averages of the headline (N = 2000 x 32) with the panel launch forced into each of its shapes:
#   bash tools/panel_shape_probe.sh <tag> "0 1 2 3"      ("panel_shape": 0 by size, 1 k_panels_cad_ks, 2 k_panels_cad<1>, 3 k_panels_cad<4>)
# (profiles/r06_panel_launch.txt)
export TMPDIR=/tmp
tag=$1; shapes=${2:-"0 1 2"}
OUT=gpurun_out/ps_$tag
rm -rf $OUT; mkdir -p $OUT
for s in $shapes; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/s$s -o run -- python3 bench.py --no-cpu-baseline --no-single --option panel_shape=$s > $OUT/s$s.log 2>&1 || exit 1
  echo "== panel_shape=$s" >> $OUT/summary.txt
  python3 tools/kernel_times.py $OUT/s$s | grep "k_panels_cad\|k_solve_cad\|k_flush_rs" >> $OUT/summary.txt
  grep '^{"metric"' $OUT/s$s.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps/s', round(d['value']), 'pass ms', round(d['roofline']['avg_launch_ms'],4))" >> $OUT/summary.txt
done
cat $OUT/summary.txt
