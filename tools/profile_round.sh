# Collects what profiles/ holds for a round: kernel-trace stats (B=32, B=1) and the PMC traffic passes.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_round
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b32 -o run -- python3 bench.py --no-cpu-baseline --no-single > $OUT/stats_b32.log 2>&1
echo "stats b32 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b1 -o run -- python3 bench.py --no-cpu-baseline --no-single --trajectories 1 > $OUT/stats_b1.log 2>&1
echo "stats b1 done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_b32 -o run -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-single > $OUT/pmc_fetch_b32.log 2>&1
echo "pmc fetch b32 done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_b32 -o run -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-single > $OUT/pmc_write_b32.log 2>&1
echo "pmc write b32 done"
# (counter collection runs one kernel at a time: the chained solves' launches wait for one another across the two streams on
#  device-side counters and would run into their bounds -- the PMC passes of the small launches take the round-3 look-ahead,
#  whose hand-overs are stream events; the pass kernel and its launch shape are the same)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_b1 -o run -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-single --trajectories 1 --option chain=0 > $OUT/pmc_fetch_b1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_b1 -o run -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-single --trajectories 1 --option chain=0 > $OUT/pmc_write_b1.log 2>&1
echo "pmc b1 done"
# BASELINE config 5 (N = 8000 x 1, column-panel layout): the pass's HBM traffic
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_n8000 -o run -- python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 --option chain=0 > $OUT/pmc_fetch_n8000.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_n8000 -o run -- python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 --option chain=0 > $OUT/pmc_write_n8000.log 2>&1
echo "pmc n8000 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_n8000 -o run -- python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 --steps 100 > $OUT/stats_n8000.log 2>&1
python3 tools/kernel_times.py $OUT/stats_b32 $OUT/stats_b1 $OUT/stats_n8000 > $OUT/kernel_times.txt
python3 tools/cad_timeline.py $OUT/stats_b1 3 > $OUT/timeline_b1.txt
python3 tools/cad_timeline.py $OUT/stats_n8000 2 > $OUT/timeline_n8000.txt
python3 tools/pmc_summary.py $OUT/pmc_fetch_b32 $OUT/pmc_write_b32 $OUT/pmc_fetch_b1 $OUT/pmc_write_b1 $OUT/pmc_fetch_n8000 $OUT/pmc_write_n8000 > $OUT/pmc.txt
cat $OUT/kernel_times.txt $OUT/pmc.txt
