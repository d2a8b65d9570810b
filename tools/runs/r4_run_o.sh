#!/bin/bash
# round-4 GPU call O: streaming attention backward (N >= 560) -- parity tests, kernel-level A/B against the chunk kernels, race hunt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4o
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" > $O/pytest_attn.log 2>&1; echo "pytest(attention) rc=$?"; tail -15 $O/pytest_attn.log
PREV=tools/_build_prod_chunkbwd/libmetaenc.so timeout 900 bash tools/ab_attn_shapes.sh "32 1568 16 64" "64 592 12 64" "64 560 12 64" "32 1000 16 64" "16 3136 16 64" "64 640 12 48" 2>&1 | tee $O/ab_attn_bwd.txt
timeout 600 python tools/attn_stress.py 60 2>&1 | tee $O/stress.txt
