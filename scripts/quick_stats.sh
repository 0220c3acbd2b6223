#!/bin/bash
# one rocprofv3 --kernel-trace --stats pass over ONE stream of the bench (isolated kernel durations: nothing overlaps)
#   scripts/quick_stats.sh <tag> [bench args...]   -> gpurun_out/<tag>_kernel_stats.csv
set -e -o pipefail
TAG=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=32
CACHE=gpurun_out/${TAG}_frames.npz
python3 bench.py --steps 1 --warmup 0 --no-cpu --no-h2d --no-isolated --no-sub --streams 1 --pairs-per-gpu 256 --frames-cache $CACHE "$@" > gpurun_out/${TAG}_plain.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-h2d --no-isolated --no-sub --render-workers 1 --streams 1 --pairs-per-gpu 256 --frames-cache $CACHE --detail-out gpurun_out/${TAG}_detail.json "$@" > gpurun_out/prof_${TAG}.log 2>&1
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/prof_$TAG $CACHE
python3 - "$TAG" <<'PY'
import csv, sys
rows = list(csv.DictReader(open("gpurun_out/%s_kernel_stats.csv" % sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    import re
    m = re.search(r"(\w+_kernel|\w+Buffer\w*|elementwise\w*)", r["Name"])
    name = m.group(1) if m else r["Name"][:34]
    print("%-34s calls %4s  avg %8.1f us  min %8.1f us  per step %7.3f ms  %5.1f %%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["TotalDurationNs"]) / 8e6, 100 * float(r["TotalDurationNs"]) / tot))
print("kernel time per step of 256 pairs: %.3f ms" % (tot / 8e6))
PY
