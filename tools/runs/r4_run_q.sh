#!/bin/bash
# round-4 GPU call Q: LayerNorm affine-gradient folds of a block in one launch + four-element transposes -- tests, bench A/B against HEAD;
# streaming backward below N = 513 (arm ME_ST_BWD_MINN=400) against the mid kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4q
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py tests/test_gpu_optim.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
KEEP=$O REPS=2 bash tools/ab_bench.sh new=/tmp/cur.so head=tools/_build_prod_head/libmetaenc.so 2>&1 | tee $O/ab_bench.txt
echo "== backward: cur = mid kernel (N <= 512), prev = streaming from N = 400"
PREV=tools/_build_prod_st400/libmetaenc.so timeout 600 bash tools/ab_attn_shapes.sh "128 512 16 64" "64 512 12 64" "64 448 12 64" "128 400 12 64" 2>&1 | grep -o "^cur.*\|^prev.*" | sed 's/fwd .*bwd/bwd/' | tee $O/ab_bwd_mid.txt
