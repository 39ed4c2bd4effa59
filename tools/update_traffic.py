#!/usr/bin/env python3
"""Write profiles/pass_traffic.json (what bench.py reports as roofline.traffic) from the PMC passes of
tools/profile_round.sh: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate runs, counters in KiB.
FETCH_SIZE is doubled (gfx950 reports half of the bytes of wide coalesced reads, MI355X_MICROARCH.md section HBM);
WRITE_SIZE is exact for 16-B-per-lane streaming stores.  The entry is stamped with a hash of the kernel source
so that bench.py can tell when it has gone stale (the stamp covers every file under csrc/).

  python3 tools/update_traffic.py gpurun_out/prof_round 2000 [--tag r02]
"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pass_kernel(d, counter):
    """(name, launches, average KiB) of the covariance-pass kernel with the most launches in a PMC directory."""
    best = None
    for (name, grid, ctr), (n, avg) in summarise(d).items():
        if ctr == counter and "k_flush" in name and (best is None or n > best[1]):
            best = (name + " grid " + grid, n, avg)
    return best


def main():
    out_dir, landmarks = sys.argv[1], int(sys.argv[2])
    tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else "r02"
    n = 3 + 2 * landmarks
    sys.path.insert(0, ROOT)
    import bench                                   # the stamp bench.py checks: a hash over all of csrc/*.hip and csrc/*.h
    sha = bench.kernel_source_sha()
    path = os.path.join(ROOT, "profiles", "pass_traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    for B, suffix in ((32, "b32"), (1, "b1")):
        f = pass_kernel(os.path.join(out_dir, "pmc_fetch_" + suffix), "FETCH_SIZE")
        w = pass_kernel(os.path.join(out_dir, "pmc_write_" + suffix), "WRITE_SIZE")
        if not f or not w:
            continue
        fetch, write = f[2] * 1024.0, w[2] * 1024.0
        data[f"N{landmarks}_B{B}"] = {
            "kernel": f[0],
            "launches_averaged": [f[1], w[1]],
            "FETCH_SIZE_bytes_raw": fetch,
            "FETCH_SIZE_bytes_corrected_x2": 2.0 * fetch,
            "WRITE_SIZE_bytes": write,
            "hbm_bytes_per_launch": 2.0 * fetch + write,
            "algorithmic_bytes_per_launch": B * 8.0 * n * (n + 1),
            "kernel_source_sha256_16": sha,
            "source": f"profiles/{tag}_pmc_flush.txt: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, "
                      "tools/profile_round.sh), bench.py --steps 12 --warmup 4; FETCH_SIZE doubled per "
                      "MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads)",
        }
    json.dump(data, open(path, "w"), indent=1)
    print(json.dumps(data, indent=1))


if __name__ == "__main__":
    main()
