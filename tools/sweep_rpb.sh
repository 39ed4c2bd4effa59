set -e
for rpb in 32 64 96 128 256; do
  echo "B=1 rpb=$rpb"; python bench.py --trajectories 1 --no-cpu-baseline --no-single --option pass_rows_per_block=$rpb | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"
done
for rpb in 128 256 512 1024; do
  echo "B=32 rpb=$rpb"; python bench.py --no-cpu-baseline --no-single --option pass_rows_per_block=$rpb | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"
done
