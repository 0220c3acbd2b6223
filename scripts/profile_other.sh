#!/bin/bash
# rocprofv3 passes over the OTHER configurations' bench (scripts/bench_other_configs.py), run on the GPU box from the repo root:
#   scripts/profile_other.sh <tag> C3            -> gpurun_out/<tag>_{kernel_stats,pmc_hbm_per_kernel,sq_per_kernel}.csv
#   scripts/profile_other.sh <tag> C5 EPNP|KNEIP
# The same pass structure as scripts/profile_bench.sh: --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; --pmc SQ_*
# (PMC passes never share a run with tracing).  The inputs are rendered ONCE (single process: no fork under the profiler's
# preloaded tool) into gpurun_out/<tag>_inputs.npz and reused by every pass.  Copy the summaries into profiles/roundN/.
set -e -o pipefail
TAG=${1:-run}
CFG=${2:-C3}
ALGO=${3:-EPNP}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=32   # what the package asks for on import: under the profiler HIP is initialised before Python starts
if [ "$CFG" = "C3" ]; then
  FRAMES=${FRAMES:-48}
  ARGS="--only C3 --frames $FRAMES --render-workers 1 --steps 4 --cache gpurun_out/${TAG}_inputs.npz"
  UNITS=$((FRAMES / 3))      # frames one launch covers (three streams)
else
  PAIRS=${PAIRS:-32}
  ARGS="--only C5 --pairs $PAIRS --c5-algo $ALGO --render-workers 1 --steps 4 --cache gpurun_out/${TAG}_inputs.npz"
  UNITS=$PAIRS               # pairs one launch covers (one one-call batch per stream)
fi
python3 scripts/bench_other_configs.py $ARGS > gpurun_out/${TAG}_plain.json 2> gpurun_out/${TAG}_plain.err   # renders + caches the inputs
echo "plain run done"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG --output-format csv -- python3 scripts/bench_other_configs.py $ARGS > gpurun_out/prof_${TAG}.log 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_${TAG}_fetch --output-format csv -- python3 scripts/bench_other_configs.py $ARGS > gpurun_out/pmc_${TAG}_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_${TAG}_write --output-format csv -- python3 scripts/bench_other_configs.py $ARGS > gpurun_out/pmc_${TAG}_write.log 2>&1
echo "write pass done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES -d gpurun_out/pmc_${TAG}_sq --output-format csv -- python3 scripts/bench_other_configs.py $ARGS > gpurun_out/pmc_${TAG}_sq.log 2>&1
echo "sq pass done"
python3 scripts/summarize_sq.py gpurun_out/pmc_${TAG}_sq > gpurun_out/${TAG}_sq_per_kernel.csv
python3 scripts/summarize_pmc.py gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write $UNITS > gpurun_out/${TAG}_pmc_hbm_per_kernel.csv
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -f gpurun_out/${TAG}_inputs.npz
