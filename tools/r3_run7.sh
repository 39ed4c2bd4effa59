set -e
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_cadence.py -x -q 2>&1 | tail -4
bash tools/r3_trace_b1.sh | head -9
for B in 1 2 4; do
  for L in 1 0; do
    python3 bench.py --no-cpu-baseline --no-single --trajectories $B --option lookahead=$L 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=2000 B=$B lookahead=$L  %.0f steps/s  pass %.1f us' % (d['value'], d['roofline']['avg_launch_ms']*1e3))"
  done
done
