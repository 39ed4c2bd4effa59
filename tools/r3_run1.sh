# round 3, GPU run 1: the whole GPU suite, then the pass with the three store forms
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/r3a/pytest.log | tail -15
for v in "" old tied; do
  EKFSLAM_HIP_VARIANT=$v python3 tools/flush_time.py 2>/dev/null | tee -a gpurun_out/r3a/store_forms.txt
done
for v in "" old tied; do
  EKFSLAM_HIP_VARIANT=$v python3 tools/flush_time.py 2>/dev/null | tee -a gpurun_out/r3a/store_forms.txt
done
