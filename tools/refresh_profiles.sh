#!/bin/bash
# After `bash tools/profile_round.sh` on the GPU box (outputs merged back under gpurun_out/prof_round): copy the summaries the
# judge reads into profiles/ and re-stamp profiles/pass_traffic.json with the current kernel sources.   bash tools/refresh_profiles.sh r04
set -e
TAG=${1:-r04}
P=gpurun_out/prof_round
cd "$(dirname "$0")/.."
cp $P/stats_b32/run_kernel_stats.csv profiles/${TAG}_rocprofv3_kernel_stats_b32.csv
cp $P/stats_b1/run_kernel_stats.csv profiles/${TAG}_rocprofv3_kernel_stats_b1.csv
cp $P/stats_n8000/run_kernel_stats.csv profiles/${TAG}_rocprofv3_kernel_stats_n8000.csv
cp $P/kernel_times.txt profiles/${TAG}_kernel_times.txt
cp $P/pmc.txt profiles/${TAG}_pmc_flush.txt
python3 tools/update_traffic.py $P 2000 --tag $TAG > /dev/null
python3 - <<PY
import json, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench
from pmc_summary import summarise
path = 'profiles/pass_traffic.json'
d = json.load(open(path))
def pick(dirn, ctr):
    best = None
    for (name, grid, c), (n, avg) in summarise(dirn).items():
        if c == ctr and 'k_flush' in name and (best is None or n > best[1]):
            best = (name + ' grid ' + grid, n, avg)
    return best
f, w = pick('$P/pmc_fetch_n8000', 'FETCH_SIZE'), pick('$P/pmc_write_n8000', 'WRITE_SIZE')
n = 16003
d['N8000_B1'] = {"kernel": f[0], "launches_averaged": [f[1], w[1]], "FETCH_SIZE_bytes_raw": f[2] * 1024,
                 "FETCH_SIZE_bytes_corrected_x2": 2 * f[2] * 1024, "WRITE_SIZE_bytes": w[2] * 1024,
                 "hbm_bytes_per_launch": 2 * f[2] * 1024 + w[2] * 1024, "algorithmic_bytes_per_launch": 8.0 * n * (n + 1),
                 "kernel_source_sha256_16": bench.kernel_source_sha(),
                 "source": "profiles/${TAG}_pmc_flush.txt: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/profile_round.sh), "
                           "tools/flush_time.py --landmarks 8000 --trajectories 1; FETCH_SIZE doubled per MI355X_MICROARCH.md"}
json.dump(d, open(path, 'w'), indent=1)
for k, v in d.items():
    print(k, v['kernel'], round(v['hbm_bytes_per_launch'] / v['algorithmic_bytes_per_launch'], 3), v['kernel_source_sha256_16'])
PY
grep -h "k_flush\|k_solve_cad\|k_panels_cad" profiles/${TAG}_kernel_times.txt | head -12
