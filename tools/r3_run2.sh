set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r3b
timeout -k 10 600 python -m pytest tests/test_gpu_cadence.py -x -q 2>&1 | tee gpurun_out/r3b/pytest_cad.log | tail -30
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/r3b/pytest.log | tail -8
python bench.py --no-cpu-baseline > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err
cat gpurun_out/r3b/bench.json
python bench.py --no-cpu-baseline --no-single --option fused_cadence=0 > gpurun_out/r3b/bench_nofuse.json 2>> gpurun_out/r3b/bench.err
cat gpurun_out/r3b/bench_nofuse.json
