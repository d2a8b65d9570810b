#!/bin/bash
# round-4 GPU call B: bench-level A/B of the nt-store arms; plain (no time stamps) sustained gemm_dev loops of the same arms;
# the GPU suite (new tests)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4b
mkdir -p $O
cd $R
REPS=3 bash tools/ab_bench.sh base=/tmp/cur.so cnt=tools/_build_prod_cnt/libmetaenc.so cntc=tools/_build_prod_cntc/libmetaenc.so 2>&1 | tee $O/ab_bench.txt
CASES="g3:50432:2304:768:0 g3:50432:768:768:2 g3:50432:3072:768:1 g3:50432:3072:768:7 g3:50432:768:3072:2 g3:50432:3072:768:6"
for round in 1 2; do
for V in "" _cnt _csc1; do
  B=$R/tools/_build$V
  echo "== arm ${V:-_base} (pass $round)"
  timeout 300 $B/gemm_dev --iters 30 --power 0.7 $CASES 2>&1 | tee $O/gd${V:-_base}_$round.txt | grep -E "TF/s|power:"
done
done
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
