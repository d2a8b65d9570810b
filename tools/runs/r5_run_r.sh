#!/bin/bash
# round-5 GPU call R: PMC passes of the bench command on the FINAL kernel sources (profiles/r05_pmc_*.json carry their hash), + the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ME_WGRAD_OVERLAP=0 MODE=train bash tools/pmc_bench.sh 2>&1 | tail -4
MODE=fwd bash tools/pmc_bench.sh 2>&1 | tail -4
ROUND=r05 python tools/pmc_summary.py > gpurun_out/pmc_summary_r05.txt 2>&1
mkdir -p gpurun_out/r5r; cp profiles/r05_pmc_train.json profiles/r05_pmc_fwd.json gpurun_out/r5r/
timeout 900 python bench.py > gpurun_out/r5r/bench_default.json 2> gpurun_out/r5r/bench_default.err
python -c "import json; j=json.load(open('gpurun_out/r5r/bench_default.json')); print(j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['traffic'], str(j['roofline']['traffic_source'])[:60], j['fwd']['ms_per_step'], j['fwd']['mfma_frac'], j['fwd']['roofline']['traffic'])"
