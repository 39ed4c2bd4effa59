#!/usr/bin/env python3
"""Timeline of the cadence kernels in a rocprofv3 --kernel-trace CSV (single trajectory: who waits for whom):
   python3 tools/cad_timeline.py <dir with *_kernel_trace.csv> [cadences to print, default 3] [solve grid x, default any]
Prints, for a window in the middle of the longest run of solve launches, every kernel with its start relative to the
window's first solve, its duration and its queue; then the period (solve start to solve start) over the whole run and the
time per cadence each kernel accounts for."""
import collections
import csv
import glob
import statistics
import sys


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("ekf::", "")
    return n.split("<")[0] + ("<" + n.split("<")[1] if "<" in n else "")


def main():
    d = sys.argv[1]
    ncad = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    files = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True))
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"),
                  r["Grid_Size_X"]) for r in rows), key=lambda e: e[0])
    solves = [i for i, e in enumerate(ev) if e[2].startswith("k_solve_cad")]
    if len(solves) < 10:
        print("fewer than 10 solve launches in the trace")
        return
    # the longest run of solves whose spacing stays below 1 ms (one timed region)
    runs, cur = [], [solves[0]]
    for a, b in zip(solves, solves[1:]):
        if ev[b][0] - ev[a][0] < 1_000_000:
            cur.append(b)
        else:
            runs.append(cur)
            cur = [b]
    runs.append(cur)
    run = max(runs, key=len)
    periods = [(ev[b][0] - ev[a][0]) / 1e3 for a, b in zip(run, run[1:])]
    mid = len(run) // 2
    i0, i1 = run[mid], run[min(mid + ncad, len(run) - 1)]
    t0 = ev[i0][0]
    print(f"== {d}: {len(run)} cadences in the run; window of {ncad} from cadence {mid}")
    for e in ev[i0:i1 + 1]:
        print(f"  {e[2]:34s} q{e[3]:>3s} grid {e[4]:>7s}  start {(e[0] - t0) / 1e3:8.1f} us  dur {(e[1] - e[0]) / 1e3:7.1f}  end {(e[1] - t0) / 1e3:8.1f}")
    print(f"period (solve start to solve start): median {statistics.median(periods):.1f} us, mean {sum(periods) / len(periods):.1f}, "
          f"min {min(periods):.1f}, max {max(periods):.1f}")
    per = collections.defaultdict(float)
    for e in ev[run[0]:run[-1]]:
        per[e[2]] += (e[1] - e[0]) / 1e3
    for k, v in sorted(per.items(), key=lambda kv: -kv[1]):
        print(f"  {k:34s} {v / (len(run) - 1):7.1f} us per cadence")


if __name__ == "__main__":
    main()
