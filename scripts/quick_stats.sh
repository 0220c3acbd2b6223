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
    name = r["Name"].split("(")[0].split("::")[-1][:34]
    print("%-36s calls %4s  avg %8.1f us  min %8.1f us  %5.1f %%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
