#!/bin/bash
# round-5 GPU call C: full GPU suite on the new plans + EPI 6 / 7 timing at the reference's row counts + refshapes (json)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5c
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25 | tee $O/pytest.txt
G=tools/_build/gemm_dev
for M in 1576 3072 6304; do
  timeout 300 $G --iters 50 --check auto:$M:3072:768:1 auto:$M:3072:768:7 auto:$M:3072:768:6 auto:$M:768:3072:6
done 2>&1 | tee $O/epi67.txt
timeout 600 python tools/refshapes.py --quick --dtypes bf16 --out $O/refshapes.json 2>/dev/null | tee $O/refshapes.txt
timeout 300 python tools/graph_latency.py --batches 1,2,4,8,16 2>/dev/null | tee $O/latency.txt
