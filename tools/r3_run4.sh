set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r3d
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/r3d/pytest.log | tail -8
python bench.py --no-cpu-baseline > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r3d/bench.json'))
print('value', d['value'], 'frac', d['roofline']['frac'], 'pass ms', d['roofline']['avg_launch_ms'])
print('single', d['single_trajectory'])
print('config5', json.dumps(d['config5'], indent=1))
print('m1', d['obs_1_per_step'])
"
