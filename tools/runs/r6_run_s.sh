#!/bin/bash
# round 6, call S: the P / dS arithmetic of the attention backward kernels on pairs (r16_pds4: v_pk_fma_f32 / v_pk_mul_f32) -- attention tests
# (incl. the new 32-key edge shapes), same-box A/B against -DME_R16_PK=0 on the attention entry points and on the config-2 train step
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -k "attention or attn" > $O/tests_attn.txt 2>&1; echo "attn rc=$?"; tail -3 $O/tests_attn.txt
PREV=tools/_build_prod_r16nopk/libmetaenc.so bash tools/ab_attn_shapes.sh "256 197 12 64" "32 1568 16 64" "128 592 12 64" "128 512 16 64" "64 257 12 64" > $O/ab.txt 2>&1
cat $O/ab.txt
REPS=3 ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-fwd-leg" bash tools/ab_bench.sh pk=metatransformer_amd/libmetaenc.so nopk=tools/_build_prod_r16nopk/libmetaenc.so > $O/ab_bench.txt 2>&1
cat $O/ab_bench.txt
