#!/bin/bash
# round 6, call N: gelu' in eight bits, second form (fma + v_cvt_pk_u8_f32 pack): its tests, same-box A/B against -DME_NO_GG8=1, then a serial
# kernel trace of either arm (per-kernel averages of the two launches that changed)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6n
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k "eight" > $O/tests_gg8.txt 2>&1; echo "gg8 rc=$?"; tail -4 $O/tests_gg8.txt
REPS=3 ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-fwd-leg" bash tools/ab_bench.sh gg8=metatransformer_amd/libmetaenc.so bf16=tools/_build_prod_nogg8/libmetaenc.so > $O/ab.txt 2>&1
cat $O/ab.txt
cp metatransformer_amd/libmetaenc.so /tmp/cur.so
for arm in gg8 bf16; do
  [ $arm = bf16 ] && cp tools/_build_prod_nogg8/libmetaenc.so metatransformer_amd/libmetaenc.so
  ME_WGRAD_OVERLAP=0 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$arm -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fwd-leg > $O/tr_$arm.json 2> $O/tr_$arm.err
  f=$(find $O/tr_$arm -name '*kernel_stats.csv' | head -1)
  echo "== $arm"; head -8 $f | cut -c1-150
  cp $f $O/stats_$arm.csv; rm -rf $O/tr_$arm
done
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
