run() { echo "== $*"; python bench.py --no-cpu-baseline --no-single "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('steps/s %.0f  ms/step %.4f  flush %.1f us x %.1f steps  frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_ms']*1e3, r['steps_per_launch'], r['frac']))"; }
for o in "" "--option pass_rows_per_block=96"; do
run --landmarks 2000 --trajectories 1 $o
run --landmarks 2000 --trajectories 2 $o
run --landmarks 500 --trajectories 1 $o
run --landmarks 500 --trajectories 8 $o
run --landmarks 500 --trajectories 32 $o
run --landmarks 1000 --trajectories 4 $o
done
for b in 2 3 4 6; do run --landmarks 2000 --trajectories $b; done
run --landmarks 8000 --trajectories 1 --steps 60 --warmup 8
