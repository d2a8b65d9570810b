#!/bin/bash
# round-4 GPU call S: row-operand slabs in flight in a whole tile's epilogue, 6 (base) against 4 (ah4: the count the statistics and the
# 128-row forms use) -- residual (2), residual + statistics (8) and x-row-operand (6) kernels, sustained loops with power
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s
mkdir -p $O
cd $R
CASES="g3:50432:768:768:2 g3:50432:768:768:8 g3:50432:768:3072:2 g3:50432:768:3072:8 g3:50432:3072:768:6"
for round in 1 2; do
  echo "== base (pass $round)"; timeout 300 tools/_build/gemm_dev --check --iters 30 --power 0.7 $CASES 2>&1 | tee $O/gd_base_$round.txt | grep -E "TF/s|power:"
  echo "== ah4 (pass $round)"; timeout 300 tools/_build_ah4/gemm_dev --check --iters 30 --power 0.7 $CASES 2>&1 | tee $O/gd_ah4_$round.txt | grep -E "TF/s|power:"
done
