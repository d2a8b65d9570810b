#!/bin/bash
# round-4 GPU call R: AdamW with four-element lanes, LayerNorm affine fold v2 (new), + 128-row items in the x-row-operand kernel (hi6)
# against HEAD -- optimizer / LayerNorm / encoder tests on the new build, then bench A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4r
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_optim.py tests/test_gpu_encoder.py tests/test_gpu_ops.py -m gpu -q -x -k "not attention" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
KEEP=$O REPS=2 bash tools/ab_bench.sh new=/tmp/cur.so hi6=tools/_build_prod_hi6/libmetaenc.so head=tools/_build_prod_head/libmetaenc.so 2>&1 | tee $O/ab_bench.txt
cp tools/_build_prod_hi6/libmetaenc.so metatransformer_amd/libmetaenc.so
timeout 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_ops.py -m gpu -q -x -k "gemm or backward or parity" > $O/pytest_hi6.log 2>&1; echo "pytest(hi6) rc=$?"; tail -4 $O/pytest_hi6.log
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
