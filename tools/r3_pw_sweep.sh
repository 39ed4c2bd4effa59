set -e
export TMPDIR=/tmp
for PW in 0 240 224 208; do
  for LA in 1 0; do
    python3 bench.py --no-cpu-baseline --no-single --option pass_workgroups=$PW --option lookahead=$LA 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=2000 B=32 pass_workgroups=$PW lookahead=$LA  %.0f steps/s  pass %.1f us' % (d['value'], d['roofline']['avg_launch_ms']*1e3))"
  done
done
