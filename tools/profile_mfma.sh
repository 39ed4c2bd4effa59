# MFMA-busy evidence for the two kernels that use the fp64 matrix cores (own PMC pass, no tracing domains)
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_mfma
rm -rf $OUT; mkdir -p $OUT
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64"
rocprofv3 --pmc $C --output-format csv -d $OUT/dense -o run -- python3 tools/dense_time.py > $OUT/dense.log 2>&1
echo "dense done"
rocprofv3 --pmc $C --output-format csv -d $OUT/flush -o run -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-single > $OUT/flush.log 2>&1
echo "flush done"
rocprofv3 --pmc $C --output-format csv -d $OUT/flush_n8000 -o run -- python3 -W ignore tools/flush_time.py --landmarks 8000 --trajectories 1 --option chain=0 > $OUT/flush_n8000.log 2>&1
echo "flush n8000 done"
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/prof_mfma/dense", "gpurun_out/prof_mfma/flush", "gpurun_out/prof_mfma/flush_n8000"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "gemm" in name or "flush" in name:
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for name, c in agg.items():
            m = {k: sum(v) / len(v) for k, v in c.items()}
            busy, gui = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), m.get("GRBM_GUI_ACTIVE", 1)
            print(f"{name:40s} launches {len(c['GRBM_GUI_ACTIVE']):3d}  " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(m.items())))
            # GRBM_GUI_ACTIVE arrives summed over the 8 XCDs; MFMA_BUSY is summed over the 1024 SIMDs (64 cycles per
            # v_mfma_f64_16x16x4): MfmaUtil = busy / (cycles of one XCD * 1024)
            print(f"{'':40s} MfmaUtil = MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs) = {100 * busy / (gui / 8 * 1024):.1f} %   "
                  f"MFMA flop per launch = {m.get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0) * 512:.4g}   kernel cycles = {gui / 8:.4g}")
PY
