#!/bin/bash
# round-5 GPU call L: the driver's bench command on the final tree (roofline.traffic from the r05 PMC set)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5l
mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
python -c "import json; j=json.load(open('$O/bench_default.json')); print(j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['traffic'], j['fwd']['ms_per_step'], j['fwd']['mfma_frac'], j['fwd']['roofline']['traffic'], j['cpu_baseline']['value'])"
