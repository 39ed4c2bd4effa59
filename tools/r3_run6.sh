set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r3f
timeout -k 10 600 python -m pytest tests/test_gpu_cadence.py -x -q 2>&1 | tee gpurun_out/r3f/pytest_cad.log | tail -15
for B in 1 2 4 8; do
  for L in 1 0; do
    python3 bench.py --no-cpu-baseline --no-single --trajectories $B --option lookahead=$L 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=2000 B=$B lookahead=$L  %.0f steps/s  pass %.1f us' % (d['value'], d['roofline']['avg_launch_ms']*1e3))"
  done
done
for cfg in "--landmarks 500 --trajectories 1" "--landmarks 500 --trajectories 8" "--landmarks 20 --trajectories 1"; do
  for L in 1 0; do
    python3 bench.py --no-cpu-baseline --no-single $cfg --option lookahead=$L 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg lookahead=$L  %.0f steps/s  pass %.1f us' % (d['value'], d['roofline']['avg_launch_ms']*1e3))"
  done
done
