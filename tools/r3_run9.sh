set -e
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_cadence.py -x -q 2>&1 | tail -3
bash tools/r3_variants.sh default 2>&1 | grep "variant\|cad"
for B in 1 8 16 32; do
  python3 bench.py --no-cpu-baseline --no-single --trajectories $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=2000 B=$B  %.0f steps/s  pass %.1f us' % (d['value'], d['roofline']['avg_launch_ms']*1e3))"
done
python3 bench.py --no-cpu-baseline --no-single --landmarks 500 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=500 B=32  %.0f steps/s' % d['value'])"
python3 bench.py --no-cpu-baseline --no-single --obs 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=2000 B=32 m=1 %.0f steps/s' % d['value'])"
