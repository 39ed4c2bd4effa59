# steps/s with and without the fused cadence over the batch size (N=2000, m=8) and for m=1, N=500
for B in 1 2 4 8 16 24 32; do
  for F in 1 0; do
    python3 bench.py --no-cpu-baseline --no-single --trajectories $B --option fused_cadence=$F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=2000 B=$B fused=$F  %.0f steps/s  pass %.1f us' % (d['value'], d['roofline']['avg_launch_ms']*1e3))"
  done
done
for cfg in "--landmarks 500 --trajectories 1" "--landmarks 500 --trajectories 32" "--landmarks 20 --trajectories 1" "--landmarks 8000 --trajectories 1 --steps 100"; do
  for F in 1 0; do
    python3 bench.py --no-cpu-baseline --no-single $cfg --option fused_cadence=$F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg fused=$F  %.0f steps/s  pass %.1f us' % (d['value'], d['roofline']['avg_launch_ms']*1e3))"
  done
done
