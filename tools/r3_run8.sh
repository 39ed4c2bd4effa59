set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r3g
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/r3g/pytest.log | tail -6
bash tools/r3_fuse_sweep.sh 2>&1 | tee gpurun_out/r3g/fuse_sweep.txt
