set -e
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_cadence.py -x -q 2>&1 | tail -5
bash tools/r3_variants.sh default 2>&1 | grep "variant\|cad"
for B in 1 32; do
  for F in 5 2; do
    python3 bench.py --no-cpu-baseline --no-single --trajectories $B --option solve_form=$F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=2000 B=$B solve_form=$F  %.0f steps/s  pass %.1f us' % (d['value'], d['roofline']['avg_launch_ms']*1e3))"
  done
done
for cfg in "--landmarks 500 --trajectories 1" "--landmarks 20 --trajectories 1"; do
  for F in 5 2; do
    python3 bench.py --no-cpu-baseline --no-single $cfg --option solve_form=$F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg solve_form=$F  %.0f steps/s' % d['value'])"
  done
done
