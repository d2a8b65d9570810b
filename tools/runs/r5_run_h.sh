#!/bin/bash
# round-5 GPU call H: x3 attention v2 (two query tiles per wave): tests + timing
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5h
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -k "x3 or 3xbf16" 2>&1 | tail -15 | tee $O/pytest_x3.txt
timeout 300 python tools/attn_x3_time.py 2>&1 | tee $O/attn_x3_time.txt
timeout 300 python tools/attn_x3_time.py 32 1568 16 2>&1 | tee -a $O/attn_x3_time.txt
