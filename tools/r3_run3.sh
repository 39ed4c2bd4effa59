set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r3c
timeout -k 10 600 python -m pytest tests/test_gpu_cadence.py -x -q 2>&1 | tee gpurun_out/r3c/pytest_cad.log | tail -5
EKFSLAM_HIP_VARIANT=stamps python3 tools/cad_stamps.py 2>/dev/null | tee gpurun_out/r3c/cad_stamps.txt
bash tools/r3_kt.sh 2>&1 | tee gpurun_out/r3c/kt.txt
