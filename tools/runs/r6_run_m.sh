#!/bin/bash
# round 6, call M: gelu' in eight bits (ME_GG8) -- its GEMM tests, the training parity tests, then a same-box A/B against the build that keeps
# the bf16 factor (-DME_NO_GG8=1)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6m
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k "eight" > $O/tests_gg8.txt 2>&1; echo "gg8 rc=$?"; tail -6 $O/tests_gg8.txt
timeout 2400 python -m pytest tests -m gpu -x -q -k "backward or train or bwd or saved or grad" > $O/tests_train.txt 2>&1; echo "train rc=$?"; tail -6 $O/tests_train.txt
REPS=3 bash tools/ab_bench.sh gg8=metatransformer_amd/libmetaenc.so bf16=tools/_build_prod_nogg8/libmetaenc.so > $O/ab.txt 2>&1
cat $O/ab.txt
