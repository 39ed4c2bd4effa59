for fe in 1 2 3 4 5; do
  python3 tools/flush_time.py --landmarks 8000 --trajectories 1 --steps 40 --option flush_every=$fe 2>/dev/null
done
for fe in 1 5; do
  python3 tools/flush_time.py --landmarks 8000 --trajectories 1 --steps 40 --option flush_every=$fe --option pass_kernel=0 2>/dev/null
done
python3 tools/flush_time.py --landmarks 8000 --trajectories 2 --steps 40 2>/dev/null
python3 tools/flush_time.py --landmarks 8000 --trajectories 2 --steps 40 --option pass_kernel=0 2>/dev/null
python3 tools/flush_time.py --landmarks 4000 --trajectories 4 --steps 40 2>/dev/null
python3 tools/flush_time.py --landmarks 4000 --trajectories 4 --steps 40 --option pass_kernel=0 2>/dev/null
python3 tools/flush_time.py --landmarks 4000 --trajectories 8 --steps 40 2>/dev/null
python3 tools/flush_time.py --landmarks 4000 --trajectories 8 --steps 40 --option pass_kernel=0 2>/dev/null
