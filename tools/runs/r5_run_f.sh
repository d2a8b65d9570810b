#!/bin/bash
# round-5 GPU call F: tightened attention / per-layer tests, then the round profile (tools/prof_round.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5f
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention_fwd or attention_bwd" 2>&1 | tail -25 | tee $O/pytest_attn.txt
timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -k "layer_by_layer" 2>&1 | tail -25 | tee $O/pytest_layer.txt
bash tools/prof_round.sh 2>&1 | tail -40 | tee $O/prof_round.txt
