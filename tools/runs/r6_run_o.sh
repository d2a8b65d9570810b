#!/bin/bash
# round 6, call O: streaming attention backward with 32 rows per wave -- the attention tests, then same-box A/B (attention entry points alone)
# against the 16-row kernels (-DME_ST_BWD_ROWS=16) at configs 3 / 4 / 5 and around
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6o
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -k "attention or attn" > $O/tests_attn.txt 2>&1; echo "attn rc=$?"; tail -3 $O/tests_attn.txt
PREV=tools/_build_prod_rows16/libmetaenc.so bash tools/ab_attn_shapes.sh "32 1568 16 64" "128 592 12 64" "64 1000 12 64" "128 520 16 64" "16 3136 16 64" "64 1568 12 64" > $O/ab.txt 2>&1
cat $O/ab.txt
bash tools/runs/r6_run_p.sh
