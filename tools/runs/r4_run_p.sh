#!/bin/bash
# round-4 GPU call P: (1) fp8 attention with the exponential taken as a bit pattern (-DF8_FAST_EXP=1) against the shipped fp8 kernel and
# the bf16 streaming forward: kernel times by rocprofv3, accuracy by the fp8 tests; (2) streaming backward at N = 512 against the mid kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4p
mkdir -p $O
cd $R
export TMPDIR=/tmp
cp metatransformer_amd/libmetaenc.so /tmp/cur.so
for arm in base fastexp; do
  [ $arm = base ] && cp /tmp/cur.so metatransformer_amd/libmetaenc.so || cp tools/_build_prod_fastexp/libmetaenc.so metatransformer_amd/libmetaenc.so
  echo "== $arm"
  timeout 300 python tools/attn_fp8_bench.py 2>/dev/null | tee $O/fp8_bench_$arm.json
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$arm -o t -- python $R/tools/attn_fp8_bench.py > /dev/null 2>&1)
  python - $arm <<'PY' | tee $O/fp8_kernels_$arm.txt
import csv,glob,sys
arm=sys.argv[1]
f=glob.glob(f'/tmp/prof_{arm}/**/*kernel_stats.csv',recursive=True)
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    if 'f8' in n or 'fp8' in n or 'stream16' in n:
        print(f"{arm:8s} {n[:70]:70s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
  timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "fp8" 2>&1 | tail -12 | tee $O/fp8_tests_$arm.txt
done
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
echo "== backward at N = 512: cur = mid kernel, prev = streaming (ME_ST_BWD_MINN=500)"
PREV=tools/_build_prod_st512/libmetaenc.so timeout 600 bash tools/ab_attn_shapes.sh "128 512 16 64" "64 512 12 64" "128 520 12 64" "128 544 12 64" 2>&1 | tee $O/ab_bwd_512.txt
