#!/bin/bash
# round 6, call X: the dQ phase of the streaming backward on 32-query waves (12 waves, staged steps; -DME_ST32_DQ=1) against the shipped 16-row
# kernel: attention tests on the arm, per-kernel times, entry-point A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6x
mkdir -p $O
cd $R
cp metatransformer_amd/libmetaenc.so /tmp/cur.so
cp tools/_build_prod_dq32/libmetaenc.so metatransformer_amd/libmetaenc.so
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -k "attention or attn" > $O/tests_attn.txt 2>&1; echo "attn (dq32 arm) rc=$?"; tail -3 $O/tests_attn.txt
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
ARMS="dq32" bash tools/runs/r6_run_p.sh 2>&1 | tee $O/kernels.txt
