for g in 1 2 4 8; do
  python3 tools/flush_time.py --landmarks 8000 --trajectories 1 --steps 40 --option pass_share_group=$g 2>/dev/null
done
for g in 1 4; do
  python3 tools/flush_time.py --landmarks 8000 --trajectories 1 --steps 40 --option pass_share_group=$g --option flush_every=1 2>/dev/null
  python3 tools/flush_time.py --landmarks 4000 --trajectories 4 --steps 40 --option pass_share_group=$g 2>/dev/null
done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "equal_static_shares" 2>&1 | tail -3
