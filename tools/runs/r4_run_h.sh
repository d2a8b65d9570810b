#!/bin/bash
# round-4 GPU call H: in-kernel wgrad fold (tests, bench A/B against the separate fold launch); small-batch latency with round 1's
# fold loop; 128-row items in the residual kernel, kernel level, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4h
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py -m gpu -q -x -k "gemm or backward or flat_params" > $O/pytest_new.log 2>&1; echo "pytest(new) rc=$?"; tail -5 $O/pytest_new.log
P=tools/_build_prod
KEEP=$O REPS=3 ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-fwd-leg" bash tools/ab_bench.sh fold=/tmp/cur.so nofold=${P}_nofold/libmetaenc.so 2>&1 | tee $O/ab_bench.txt
cp metatransformer_amd/libmetaenc.so /tmp/head.so
for rep in 1 2; do
  timeout 300 python tools/graph_latency.py --pkg tools/_build_r1 --batches 1,8 2>&1 | grep -E "B=" | sed 's/^/r1         /' | tee -a $O/latency.txt
  timeout 300 python tools/graph_latency.py --batches 1,8 2>&1 | grep -E "B=" | sed 's/^/head       /' | tee -a $O/latency.txt
  cp ${P}_foldsimple/libmetaenc.so metatransformer_amd/libmetaenc.so
  timeout 300 python tools/graph_latency.py --batches 1,8 2>&1 | grep -E "B=" | sed 's/^/foldsimple /' | tee -a $O/latency.txt
  cp /tmp/head.so metatransformer_amd/libmetaenc.so
done
CASES="g3:50432:768:768:2 g3:50432:768:3072:2"
for round in 1 2; do
  echo "== base (pass $round)"; timeout 300 tools/_build/gemm_dev --iters 30 --power 0.8 $CASES g3:50432:768:768:8 g3:50432:768:3072:8 2>&1 | tee $O/gd_base_$round.txt | grep -E "power:" 
  echo "== hi2 (pass $round)"; timeout 300 tools/_build_hi2/gemm_dev --iters 30 --power 0.8 $CASES 2>&1 | tee $O/gd_hi2_$round.txt | grep -E "power:"
done
