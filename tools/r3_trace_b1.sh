export TMPDIR=/tmp
rm -rf gpurun_out/tr_b1; mkdir -p gpurun_out/tr_b1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_b1 -o run -- python3 bench.py --no-cpu-baseline --no-single --trajectories 1 --steps 40 --warmup 10 > gpurun_out/tr_b1/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tr_b1/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 24 kernels
t0 = int(rows[-24]["Start_Timestamp"])
for r in rows[-24:]:
    name = r["Kernel_Name"].split("(")[0].replace("void ekf::", "")[:28]
    print(f"{name:30s} start {int(r['Start_Timestamp'])-t0:8d} ns  end {int(r['End_Timestamp'])-t0:8d} ns  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f} us  queue {r.get('Queue_Id','?')} stream {r.get('Stream_Id','?')}")
PY
