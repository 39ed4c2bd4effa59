# Where the waves of the covariance pass spend their cycles: SQ counters of one PMC pass (no tracing domains).
#   bash tools/pmc_pass.sh [flush_time.py arguments]
set -e
export TMPDIR=/tmp
OUT=gpurun_out/pmc_pass
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS \
  --output-format csv -d $OUT/a -o run -- python3 ${PMC_PROG:-tools/flush_time.py} "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_LDS \
  --output-format csv -d $OUT/b -o run -- python3 ${PMC_PROG:-tools/flush_time.py} "$@" > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_pass/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if any(k in name for k in ("k_flush", "k_panels", "k_solve", "k_gemm")):
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    print(name, "launches", len(next(iter(c.values()))))
    for k in sorted(m):
        print(f"   {k:32s} {m[k]:16.4g}")
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        print(f"   kernel cycles (GUI_ACTIVE/8) = {cyc:.4g};  MfmaUtil = {100 * m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024):.1f} % of 1024 SIMDs")
    if "SQ_WAVE_CYCLES" in m:
        w = m["SQ_WAVE_CYCLES"]
        print("   of wave cycles: wait_any %.1f %%  wait_inst_any %.1f %%  active_inst_any %.1f %%  wait_inst_lds %.1f %%" % (
            100 * m.get("SQ_WAIT_ANY", 0) / w, 100 * m.get("SQ_WAIT_INST_ANY", 0) / w, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / w,
            100 * m.get("SQ_WAIT_INST_LDS", 0) / w))
PY
