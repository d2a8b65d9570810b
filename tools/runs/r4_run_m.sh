#!/bin/bash
# round-4 GPU call M: LayerNorm kernels (backward with the residual gradient as a template parameter; forward / statistics two rows
# per wave) -- tests and bench A/B against the previous LayerNorm
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4m
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py -m gpu -q -x -k "layernorm or row_stat or folded or backward" > $O/pytest_new.log 2>&1; echo "pytest(new) rc=$?"; tail -5 $O/pytest_new.log
P=tools/_build_prod
KEEP=$O REPS=3 bash tools/ab_bench.sh newln=/tmp/cur.so prevln=${P}_prevln/libmetaenc.so 2>&1 | tee $O/ab_bench.txt
