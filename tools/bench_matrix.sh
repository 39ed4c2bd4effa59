# the configurations quoted in DESIGN.md section 4 (one line each)
run() { echo "== $*"; python bench.py --no-cpu-baseline --no-single "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('steps/s %.0f  ms/step %.4f  flush %.1f us x %.1f steps  frac %.3f mfma %.1f TF' % (d['value'], d['ms_per_step'], r['avg_launch_ms']*1e3, r['steps_per_launch'], r['frac'], r['mfma']['achieved']))"; }
run --landmarks 20 --trajectories 1
run --landmarks 20 --trajectories 32
run --landmarks 500 --trajectories 1
run --landmarks 500 --trajectories 32
run --landmarks 2000 --trajectories 1
run --landmarks 2000 --trajectories 1 --option pass_rows_per_block=96
run --landmarks 2000 --trajectories 1 --option pass_rows_per_block=64
run --landmarks 2000 --trajectories 8
run --landmarks 2000 --trajectories 32 --obs 1
run --landmarks 2000 --trajectories 32 --obs 2
run --landmarks 2000 --trajectories 32 --obs 4
run --landmarks 2000 --trajectories 32 --obs 16
run --landmarks 8000 --trajectories 1 --steps 60 --warmup 8
