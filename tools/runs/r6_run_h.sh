#!/bin/bash
# round 6, call H: column-grouped tile walk of the resident NT kernel (weights resident in the XCD's L2): GEMM tests at the shapes it takes,
# then same-box A/B of the bench (train + forward legs) against -DG3_COLGROUPS=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6h
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm" > $O/tests_ops.txt 2>&1; echo "ops rc=$?"; tail -3 $O/tests_ops.txt
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -x -q -k "partials or plans or three_plane or folded" > $O/tests_r56.txt 2>&1; echo "r56 rc=$?"; tail -3 $O/tests_r56.txt
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -k "config2 or folded or epilogue or large" > $O/tests_enc.txt 2>&1; echo "enc rc=$?"; tail -3 $O/tests_enc.txt
REPS=3 ARGS="--steps 10 --warmup 3 --no-cpu-baseline" KEEP=$O bash tools/ab_bench.sh cg=metatransformer_amd/libmetaenc.so nocg=tools/_build_prod_nocg/libmetaenc.so > $O/ab.txt 2>&1
cat $O/ab.txt
