#!/bin/bash
# round-4 GPU call V: fc1 training epilogue (GELU + saved gelu'), erf pair (base) against the two bf16-mode polynomials (gpoly)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4v
mkdir -p $O
cd $R
CASES="g3:50432:3072:768:7 g3:50432:3072:768:1"
for round in 1 2; do
  echo "== base (pass $round)"; timeout 300 tools/_build/gemm_dev --check --iters 30 --power 0.7 $CASES 2>&1 | tee $O/gd_base_$round.txt | grep -E "TF/s|power:"
  echo "== gpoly (pass $round)"; timeout 300 tools/_build_gpoly/gemm_dev --check --iters 30 --power 0.7 $CASES 2>&1 | tee $O/gd_gpoly_$round.txt | grep -E "TF/s|power:"
done
