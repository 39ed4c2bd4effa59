#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) average / median / min duration."""
import collections
import csv
import glob
import statistics
import sys


def summarise(path, min_calls=20):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        key = (name, f'{r["Grid_Size_X"]}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]}', r["Workgroup_Size_X"])
        agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = []
    for k, v in agg.items():
        if len(v) >= min_calls:
            out.append((k, len(v), sum(v) / len(v) / 1e3, statistics.median(v) / 1e3, min(v) / 1e3))
    return out


if __name__ == "__main__":
    for d in sys.argv[1:]:
        for f in sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)):
            print("==", f)
            for k, n, avg, med, mn in summarise(f):
                print(f"{k[0]:44s} grid {k[1]:>16s} wg {k[2]:>4s} calls {n:5d}  avg {avg:9.1f} us  med {med:9.1f}  min {mn:9.1f}")
