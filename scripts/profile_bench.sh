#!/bin/bash
# rocprofv3 passes over the bench command (run on the GPU box from the repo root):
#   scripts/profile_bench.sh <tag>      e.g. r1m  -> gpurun_out/prof_<tag>/, pmc_<tag>_fetch/, pmc_<tag>_write/
# 1. --kernel-trace --stats (per-kernel durations), 2./3. --pmc FETCH_SIZE / WRITE_SIZE in their own passes (they do
# not fit one pass on gfx950, and PMC passes must not be combined with API tracing).  The program itself follows
# `--` (no env/bash hop: the profiler's preloaded tool initialises the GPU first).  --render-workers 1: no fork
# under the profiler.  Then: python scripts/summarize_pmc.py ... > profiles/roundN/<tag>_pmc_hbm_per_kernel.csv
set -e -o pipefail
TAG=${1:-run}
PAIRS_PER_LAUNCH=${2:-256}   # pairs one kernel launch covers = pairs-per-gpu / streams (bench defaults: 768 / 3)
EXTRA_ARGS=${EXTRA_ARGS:-}   # e.g. EXTRA_ARGS="--ransac-solver GP3P" or "--detector ORB" for the other configurations
CACHE=gpurun_out/${TAG}_frames.npz
ARGS="--steps 10 --warmup 2 --no-cpu --no-h2d --no-isolated --no-sub --render-workers 1 --frames-cache $CACHE $EXTRA_ARGS"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=32   # what the package asks for on import: under the profiler HIP is initialised before Python starts
# render the frames ONCE, outside the profiler (forked workers), into the cache every pass below reads
python3 bench.py --steps 1 --warmup 0 --no-cpu --no-h2d --no-isolated --no-sub --frames-cache $CACHE $EXTRA_ARGS > gpurun_out/${TAG}_plain.log 2>&1
echo "frames rendered"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG --output-format csv -- python3 bench.py $ARGS --detail-out gpurun_out/${TAG}_bench_detail_under_rocprof.json > gpurun_out/prof_${TAG}_bench.log 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_${TAG}_fetch --output-format csv -- python3 bench.py $ARGS > gpurun_out/pmc_${TAG}_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_${TAG}_write --output-format csv -- python3 bench.py $ARGS > gpurun_out/pmc_${TAG}_write.log 2>&1
echo "write pass done"
if [ "${SQ_PASS:-1}" = "1" ]; then
  # 4. instruction counters (their own pass; one stream, 64 pairs per launch -- bench.py's valu_issue scales from that)
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES -d gpurun_out/pmc_${TAG}_sq --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu --no-h2d --no-isolated --no-sub --render-workers 1 --frames-cache $CACHE --streams 1 --pairs-per-gpu 64 $EXTRA_ARGS > gpurun_out/pmc_${TAG}_sq.log 2>&1
  echo "sq pass done"
  python3 scripts/summarize_sq.py gpurun_out/pmc_${TAG}_sq > gpurun_out/${TAG}_sq_per_kernel.csv
fi
python3 scripts/summarize_pmc.py gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write $PAIRS_PER_LAUNCH > gpurun_out/${TAG}_pmc_hbm_per_kernel.csv
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
grep -h '^{"metric"' gpurun_out/prof_${TAG}_bench.log > gpurun_out/${TAG}_bench_under_rocprof.json || true
# what was profiled (bench.py marks a summary stale when its own run differs): commit of the snapshot is not known on the
# GPU box (no .git there) -- the caller passes it as COMMIT=...
python3 - "$TAG" "$PAIRS_PER_LAUNCH" "$EXTRA_ARGS" <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
tag, ppl, extra = sys.argv[1], int(sys.argv[2]), sys.argv[3].split()
def opt(name, default):
    return extra[extra.index(name) + 1] if name in extra else default
meta = {"commit": os.environ.get("COMMIT") or None, "csrc_hash": bench.csrc_hash(), "detector": opt("--detector", "GFT"), "ransac_solver": opt("--ransac-solver", "P3P"),
        "pano_width": int(opt("--pano-width", 1440)), "pairs_per_launch": ppl, "streams": int(opt("--streams", 3)), "extra_args": extra}
json.dump(meta, open("gpurun_out/%s_meta.json" % tag, "w"))
PY
rm -f $CACHE
