#!/bin/bash
# memory-pipeline counters (TA / TCP / TCC) per kernel, ORB configuration, one stream of 256 pairs
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=32
CACHE=gpurun_out/pmcmem_frames.npz
A="--no-cpu --no-h2d --no-isolated --no-sub --streams 1 --pairs-per-gpu 256 --frames-cache $CACHE --detector ORB --median-win-size 0 --features-per-mask 230"
python3 bench.py --steps 1 --warmup 0 $A > /dev/null 2>&1
i=0
for set in "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d gpurun_out/pmcmem_$i --output-format csv -- python3 bench.py --steps 2 --warmup 1 --render-workers 1 $A > gpurun_out/pmcmem_$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/pmcmem_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        m = re.search(r"(\w+_kernel)", row["Kernel_Name"])
        k = m.group(1) if m else row["Kernel_Name"][:30]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = sorted({c for v in acc.values() for c in v})
print("kernel," + ",".join(names))
for k, v in acc.items():
    if k.startswith(("orb_", "unwrap_gray", "ransac_score", "match_")):
        print(k + "," + ",".join("%.4g" % (sum(v[c]) / len(v[c])) if v[c] else "" for c in names))
PY
rm -rf gpurun_out/pmcmem_[0-9] $CACHE
