#!/bin/bash
# round-4 GPU call AA: 128-row items in the x-row-operand kernel again, now with four (hi6) / two (hi6a2) row-operand slabs in flight
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4aa
mkdir -p $O
cd $R
CASES="g3:50432:3072:768:6 g3:50432:768:3072:2 g3:50432:768:768:2"
for round in 1 2; do
  for V in base hi6 hi6a2; do
    D=tools/_build_$V; [ $V = base ] && D=tools/_build
    echo "== $V (pass $round)"; timeout 300 $D/gemm_dev --check --iters 30 --power 0.7 $CASES 2>&1 | tee $O/gd_${V}_$round.txt | grep -E "TF/s|power:"
  done
done
