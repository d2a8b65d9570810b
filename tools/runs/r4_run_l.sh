#!/bin/bash
# round-4 GPU call L: the division-free fold with hoisted epilogue operands -- tests, small-batch latency vs round 1, train step vs the previous fold
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4l
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py -m gpu -q -x -k "gemm or backward or flat_params or fold or tail or small" > $O/pytest_new.log 2>&1; echo "pytest(new) rc=$?"; tail -5 $O/pytest_new.log
for rep in 1 2; do
  timeout 300 python tools/graph_latency.py --pkg tools/_build_r1 2>&1 | grep -E "B=" | sed 's/^/r1      /' | tee -a $O/latency.txt
  timeout 300 python tools/graph_latency.py 2>&1 | grep -E "B=" | sed 's/^/head    /' | tee -a $O/latency.txt
done
P=tools/_build_prod
KEEP=$O REPS=3 bash tools/ab_bench.sh newfold=/tmp/cur.so prevfold=${P}_prevfold/libmetaenc.so 2>&1 | tee $O/ab_bench.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
