#!/bin/bash
# round-3 GPU call C: -m gpu suite, resident NT kernel with / without 128-row items in the last round (g3 / g3f), bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
for F in g3 g3f; do
  timeout 300 tools/_build/gemm_dev --iters 30 --check $F:50432:768:3072:2 $F:50432:768:768:2 $F:50432:3072:768:1 $F:50432:768:2304:0 $F:50432:768:3072:0 $F:50432:3072:768:6 $F:50432:2304:768:0 > $O/gemm_$F.txt 2>&1; echo "gemm_dev $F rc=$?"
  cat $O/gemm_$F.txt
done
timeout 200 tools/_build/gemm_dev --iters 10 g3:50432:768:3072:2:8 > $O/stamps_half.txt 2>&1; head -40 $O/stamps_half.txt
timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
