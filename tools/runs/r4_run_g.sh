#!/bin/bash
# round-4 GPU call G: statistics epilogue at kernel level (epi 2 vs 8); small-batch latency after the fold fix (r1 snapshot vs HEAD)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4g
mkdir -p $O
cd $R
CASES="g3:50432:768:768:2 g3:50432:768:768:8 g3:50432:768:3072:2 g3:50432:768:3072:8"
for round in 1 2; do
  echo "== base (pass $round)"; timeout 300 tools/_build/gemm_dev --check --iters 30 --power 0.8 $CASES 2>&1 | tee $O/gd_base_$round.txt | grep -E "TF/s|power:|emit"
done
for rep in 1 2; do
  timeout 300 python tools/graph_latency.py --pkg tools/_build_r1 2>&1 | grep -E "package|B=" | tee -a $O/latency_r1.txt
  timeout 300 python tools/graph_latency.py 2>&1 | grep -E "package|B=" | tee -a $O/latency_head.txt
done
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm" 2>&1 | tail -3
