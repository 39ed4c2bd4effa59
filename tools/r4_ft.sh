#!/bin/bash
# round-4: the pass over long runs (sustained clocks) against the 20 - 40 ms runs of bench.py / flush_time.py
mkdir -p gpurun_out/r4a
O=gpurun_out/r4a/ft12.txt
: > $O
python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 --steps 40 >> $O 2>&1 || exit 1
python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 --steps 2000 >> $O 2>&1 || exit 1
python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 --steps 40 >> $O 2>&1 || exit 1
python3 -W ignore tools/flush_time.py --landmarks 2000 --trajectories 32 --steps 40 >> $O 2>&1 || exit 1
python3 -W ignore tools/flush_time.py --landmarks 2000 --trajectories 32 --steps 1500 >> $O 2>&1 || exit 1
python3 -W ignore tools/flush_time.py --landmarks 2000 --trajectories 32 --steps 40 >> $O 2>&1 || exit 1
python3 bench.py --no-cpu-baseline > gpurun_out/r4a/bench12.json 2>/dev/null
python3 -c "
import json; d=json.load(open('gpurun_out/r4a/bench12.json'))
print('bench: headline pass', d['roofline']['avg_launch_ms'], 'config5 dense pass', d['config5']['dense']['pass_avg_launch_ms'], d['config5']['dense']['value'])" >> $O
python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 --steps 100 >> $O 2>&1 || exit 1
cat $O
