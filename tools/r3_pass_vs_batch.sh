set -e
export TMPDIR=/tmp
run() { python3 bench.py --no-cpu-baseline --no-single "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-70s %.0f steps/s  pass %.1f us' % ('$*', d['value'], d['roofline']['avg_launch_ms']*1e3))"; }
for B in 32 64; do
  for RL in 16 32 48 64 80; do
    run --trajectories $B --option rank_limit=$RL
  done
done
