# Round-5 additions to tools/profile_round.sh: the headline's pass on a DENSE covariance (bench.py --leg steady_state: behind a
# full sweep of the landmarks) -- kernel-trace stats, HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes), MFMA busy -- and the
# observation-shape legs (tools/profile_shapes.sh).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_round5
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_steady -o run -- python3 bench.py --leg steady_state > $OUT/stats_steady.log 2>&1
echo "stats steady done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_steady -o run -- python3 bench.py --leg steady_state > $OUT/pmc_fetch_steady.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_steady -o run -- python3 bench.py --leg steady_state > $OUT/pmc_write_steady.log 2>&1
echo "pmc steady done"
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64"
rocprofv3 --pmc $C --output-format csv -d $OUT/mfma_steady -o run -- python3 bench.py --leg steady_state > $OUT/mfma_steady.log 2>&1
rocprofv3 --pmc $C --output-format csv -d $OUT/mfma_young -o run -- python3 bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-single > $OUT/mfma_young.log 2>&1
echo "mfma done"
python3 tools/kernel_times.py $OUT/stats_steady > $OUT/kernel_times_steady.txt
python3 tools/pmc_summary.py $OUT/pmc_fetch_steady $OUT/pmc_write_steady > $OUT/pmc_steady.txt
python3 - <<'PY' > gpurun_out/prof_round5/mfma.txt
import csv, glob, collections
for d in ("gpurun_out/prof_round5/mfma_young", "gpurun_out/prof_round5/mfma_steady"):
    print("==", d)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "k_flush_rs" in name:
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for name, c in agg.items():
            # the young-filter run: all launches; the steady run: only the timed region's launches (the last 20) count
            m = {k: sum(v[-20:]) / len(v[-20:]) for k, v in c.items()}
            busy, gui = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), m.get("GRBM_GUI_ACTIVE", 1)
            print(f"{name:40s} launches {len(c['GRBM_GUI_ACTIVE']):3d} (last 20 averaged)  " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(m.items())))
            print(f"{'':40s} MfmaUtil = MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs) = {100 * busy / (gui / 8 * 1024):.1f} %   kernel cycles = {gui / 8:.4g}")
PY
cat $OUT/kernel_times_steady.txt $OUT/pmc_steady.txt $OUT/mfma.txt
