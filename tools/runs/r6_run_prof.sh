#!/bin/bash
# round 6: the round profile (tools/prof_round.sh: rocprofv3 kernel summaries of the bench command -- train, train in serial order, forward --,
# the PMC passes, the bench lines of every BASELINE configuration, the reference's own batch shapes), folded on the box (tools/fold_profiles.sh)
# -> gpurun_out/r6prof/r06_* = what gets committed under profiles/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export ROUND=r06
bash tools/prof_round.sh > gpurun_out/prof_round.log 2>&1; echo "prof_round rc=$?"; tail -5 gpurun_out/prof_round.log
mkdir -p gpurun_out/r6prof
bash tools/fold_profiles.sh > gpurun_out/fold.log 2>&1; echo "fold rc=$?"
cp profiles/r06_*summary.txt profiles/r06_*kernel_stats.csv profiles/r06_pmc_*.json profiles/r06_bench_*.json profiles/r06_refshapes.* gpurun_out/r6prof/ 2>/dev/null
# the raw traces stay on the box (tens of MB); keep the logs
find gpurun_out/prof_r06 -name "*.csv" -delete; find gpurun_out/pmc_bench_train gpurun_out/pmc_bench_fwd -name "*.csv" -delete 2>/dev/null
ls gpurun_out/r6prof | head -40
head -12 gpurun_out/r6prof/r06_fwd_summary.txt | cut -c1-140
python - <<'PY'
import json
for f in ("base","large512_train","large1568_train","mixed","fp32_3xbf16"):
    try:
        j=json.loads(open(f"gpurun_out/r6prof/r06_bench_{f}.json").read().strip().splitlines()[-1])
        print(f, j["ms_per_step"], (j.get("fwd") or {}).get("ms_per_step"), j["roofline"]["frac"])
    except Exception as e:
        print(f, "ERR", e)
PY
