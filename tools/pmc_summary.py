#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter CSVs (FETCH_SIZE / WRITE_SIZE are in KiB)."""
import collections
import csv
import glob
import sys


def summarise(d):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[(name, r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k] = (len(v), sum(v) / len(v))
    return out


if __name__ == "__main__":
    for d in sys.argv[1:]:
        print("==", d)
        for (name, grid, ctr), (n, avg) in sorted(summarise(d).items()):
            if "ekf::" in name:
                print(f"{name:36s} grid {grid:>10s} {ctr:12s} calls {n:4d} avg {avg:14.1f} KiB = {avg * 1024 / 1e6:10.2f} MB")
