set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r3e
timeout -k 10 600 python -m pytest tests/test_gpu_cadence.py -x -q 2>&1 | tee gpurun_out/r3e/pytest_cad.log | tail -5
for B in 1 2 4 8; do
  for F in 1 0; do
    python3 bench.py --no-cpu-baseline --no-single --trajectories $B --option fused_cadence=$F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=2000 B=$B fused=$F  %.0f steps/s  pass %.1f us' % (d['value'], d['roofline']['avg_launch_ms']*1e3))"
  done
done
for cfg in "--landmarks 500 --trajectories 1" "--landmarks 500 --trajectories 8" "--landmarks 20 --trajectories 1"; do
  for F in 1 0; do
    python3 bench.py --no-cpu-baseline --no-single $cfg --option fused_cadence=$F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg fused=$F  %.0f steps/s  pass %.1f us' % (d['value'], d['roofline']['avg_launch_ms']*1e3))"
  done
done
bash tools/r3_kt.sh 2>&1 | grep "B=\|cad"
