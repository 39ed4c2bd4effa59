#!/bin/bash
# round-4 scratch: HBM/fabric fetch bytes of the pass (FETCH_SIZE, raw KiB; x2 for wide loads on gfx950) at N = 8000 x 1 against N = 2000 x 32
export TMPDIR=/tmp
OUT=gpurun_out/r4pmc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f8000_5 -o run -- python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 --option flush_every=5 > $OUT/a.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f8000_1 -o run -- python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 --option flush_every=1 > $OUT/b.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f2000_5 -o run -- python3 -W ignore tools/flush_time.py --landmarks 2000 --trajectories 32 --option flush_every=5 > $OUT/c.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f2000_1 -o run -- python3 -W ignore tools/flush_time.py --landmarks 2000 --trajectories 32 --option flush_every=1 > $OUT/d.log 2>&1 || exit 1
python3 tools/pmc_summary.py $OUT/f8000_5 $OUT/f8000_1 $OUT/f2000_5 $OUT/f2000_1 > $OUT/pmc.txt
grep flush $OUT/pmc.txt
