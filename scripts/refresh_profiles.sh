#!/bin/bash
# Build-container driver of a full profile refresh: one `gpurun` call per configuration (GPU suite + default bench, then the
# stats / FETCH / WRITE / SQ passes of the C2 headline, the ORB-detector path, GP3P, C3, C5 EPnP, C5 Kneip -- scripts/profile_bench.sh
# and scripts/profile_other.sh on the GPU box), then the summaries are copied from gpurun_out/ into profiles/<round>/ and the
# kernel tables and the static instruction mix are regenerated.  ~12 GPU-minutes.
#   ROUND=round4 PFX=r4 scripts/refresh_profiles.sh
set -u
cd "$(dirname "$0")/.."
ROUND=${ROUND:-round4}
PFX=${PFX:-r4}
C=$(git rev-parse --short HEAD)
G=/usr/local/graft/bin/gpurun
$G --timeout 1150 -- "python -m pytest tests -m gpu -x -q > gpurun_out/${PFX}_tests_final.log 2>&1; echo tests rc=\$?; tail -3 gpurun_out/${PFX}_tests_final.log; python bench.py > gpurun_out/${PFX}_bench_default.log 2>gpurun_out/${PFX}_bench_default.err; echo bench rc=\$?; tail -1 gpurun_out/${PFX}_bench_default.log | wc -c" 2>&1 | tail -6
$G --timeout 1150 -- "COMMIT=$C scripts/profile_bench.sh ${PFX}b 256 > gpurun_out/profile_${PFX}b.log 2>&1; tail -1 gpurun_out/profile_${PFX}b.log" 2>&1 | tail -3
$G --timeout 1150 -- "COMMIT=$C EXTRA_ARGS='--detector ORB --median-win-size 0 --features-per-mask 230' scripts/profile_bench.sh ${PFX}orb 256 > gpurun_out/profile_${PFX}orb.log 2>&1; tail -1 gpurun_out/profile_${PFX}orb.log" 2>&1 | tail -3
$G --timeout 1150 -- "COMMIT=$C EXTRA_ARGS='--ransac-solver GP3P' scripts/profile_bench.sh ${PFX}gp3p 256 > gpurun_out/profile_${PFX}gp3p.log 2>&1; tail -1 gpurun_out/profile_${PFX}gp3p.log" 2>&1 | tail -3
$G --timeout 1150 -- "scripts/profile_other.sh ${PFX}c3 C3 > gpurun_out/profile_${PFX}c3.log 2>&1; tail -1 gpurun_out/profile_${PFX}c3.log" 2>&1 | tail -3
$G --timeout 1150 -- "scripts/profile_other.sh ${PFX}c5epnp C5 EPNP > gpurun_out/profile_${PFX}c5epnp.log 2>&1; tail -1 gpurun_out/profile_${PFX}c5epnp.log" 2>&1 | tail -3
$G --timeout 1150 -- "scripts/profile_other.sh ${PFX}c5kneip C5 KNEIP > gpurun_out/profile_${PFX}c5kneip.log 2>&1; tail -1 gpurun_out/profile_${PFX}c5kneip.log" 2>&1 | tail -3
# ---- copy the summaries
D=profiles/$ROUND
mkdir -p $D
for t in ${PFX}b ${PFX}gp3p ${PFX}orb; do for f in kernel_stats.csv pmc_hbm_per_kernel.csv sq_per_kernel.csv bench_under_rocprof.json meta.json; do cp gpurun_out/${t}_$f $D/${t}_$f; done; done
for t in ${PFX}c3 ${PFX}c5epnp ${PFX}c5kneip; do for f in kernel_stats.csv pmc_hbm_per_kernel.csv sq_per_kernel.csv; do cp gpurun_out/${t}_$f $D/${t}_$f; done; cp gpurun_out/${t}_plain.json $D/${t}_bench.json; done
python scripts/isa_mix.py --all -o $D/${PFX}b_isa_mix.json > /dev/null 2>&1
python scripts/profile_tables.py $D > $D/KERNEL_TABLES.md
cp gpurun_out/${PFX}_bench_default.log $D/bench_default_final.json
cp gpurun_out/bench_detail.json $D/bench_default_final_detail.json 2>/dev/null || true
python - "$D" "$PFX" <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
d, pfx = sys.argv[1], sys.argv[2]
print("csrc", bench.csrc_hash(), json.load(open("%s/%sb_meta.json" % (d, pfx)))["csrc_hash"])
for t in ("b", "gp3p", "orb"):
    r = json.loads(open("%s/%s%s_bench_under_rocprof.json" % (d, pfx, t)).read().splitlines()[-1])
    print(pfx + t, round(r["value"]), r["ms_per_step"])
PY
echo "ALL DONE (a default bench run AFTER this copy reports the profile-derived fields as fresh)"
