#!/bin/bash
# one rocprofv3 --pmc SQ pass over ONE stream of 64 pairs (what bench.py's valu_issue scales from)
#   scripts/quick_sq.sh <tag> [bench args...]   -> gpurun_out/<tag>_sq_per_kernel.csv
set -e -o pipefail
TAG=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=32
CACHE=gpurun_out/${TAG}_frames.npz
python3 bench.py --steps 1 --warmup 0 --no-cpu --no-h2d --no-isolated --no-sub --streams 1 --pairs-per-gpu 64 --frames-cache $CACHE "$@" > gpurun_out/${TAG}_plain.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES -d gpurun_out/pmc_${TAG}_sq --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu --no-h2d --no-isolated --no-sub --render-workers 1 --frames-cache $CACHE --streams 1 --pairs-per-gpu 64 "$@" > gpurun_out/pmc_${TAG}_sq.log 2>&1
python3 scripts/summarize_sq.py gpurun_out/pmc_${TAG}_sq > gpurun_out/${TAG}_sq_per_kernel.csv
rm -rf gpurun_out/pmc_${TAG}_sq $CACHE
cat gpurun_out/${TAG}_sq_per_kernel.csv
