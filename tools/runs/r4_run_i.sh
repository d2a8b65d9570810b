#!/bin/bash
# round-4 GPU call I: small-batch latency with round 1's fold kernel verbatim; bench-level A/B of (a) default-policy stores for the
# residual stream, (b) 128-row items in the residual kernel, (c) both
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4i
mkdir -p $O
cd $R
P=tools/_build_prod
cp metatransformer_amd/libmetaenc.so /tmp/head.so
for rep in 1 2; do
  timeout 300 python tools/graph_latency.py --pkg tools/_build_r1 --batches 1,8 2>&1 | grep -E "B=" | sed 's/^/r1      /' | tee -a $O/latency.txt
  timeout 300 python tools/graph_latency.py --batches 1,8 2>&1 | grep -E "B=" | sed 's/^/head    /' | tee -a $O/latency.txt
  cp ${P}_foldr1/libmetaenc.so metatransformer_amd/libmetaenc.so
  timeout 300 python tools/graph_latency.py --batches 1,8 2>&1 | grep -E "B=" | sed 's/^/foldr1  /' | tee -a $O/latency.txt
  cp /tmp/head.so metatransformer_amd/libmetaenc.so
done
KEEP=$O REPS=3 bash tools/ab_bench.sh head=/tmp/cur.so cres0=${P}_cres0/libmetaenc.so hi2=${P}_hi2/libmetaenc.so hi2cres0=${P}_hi2cres0/libmetaenc.so 2>&1 | tee $O/ab_bench.txt
