#!/bin/bash
# round 6, call Q: dK / dV with 32-key waves, chunk steps software-pipelined: attention tests, then per-kernel times against the plain-order arm
# and the 16-row arm
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6q
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -k "attention or attn" > $O/tests_attn.txt 2>&1; echo "attn rc=$?"; tail -3 $O/tests_attn.txt
ARMS="st32pipe1 rows16" bash tools/runs/r6_run_p.sh 2>&1 | tee $O/kernels.txt
