#!/bin/bash
# round 6, call V: the three-product attention kernels skip tiles past N (ragged last chunk, empty waves of the last row block): parity tests,
# timing against the build without it (tools/_build_prod_x3base), fp32 bench lines of config 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6v
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py -x -q -k "x3 or 3xbf16 or three" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
for S in "256 197 12" "32 257 12" "32 96 12" "64 577 12"; do
  for arm in cur base; do
    LIB=""; [ $arm = base ] && LIB="--lib tools/_build_prod_x3base/libmetaenc.so"
    echo "== $arm $S"; python tools/attn_x3_time.py $LIB $S 2>&1 | grep -E "x3 \(out \+|bwd x3"
  done
done 2>&1 | tee $O/time.txt
cp metatransformer_amd/libmetaenc.so /tmp/cur.so
for rep in 1 2; do for arm in cur base; do
  [ $arm = base ] && cp tools/_build_prod_x3base/libmetaenc.so metatransformer_amd/libmetaenc.so || cp /tmp/cur.so metatransformer_amd/libmetaenc.so
  timeout 600 python bench.py --dtype fp32 --fp32-mode 3xbf16 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_${arm}_$rep.json 2> $O/bench.err
  python - "$O/bench_${arm}_$rep.json" $arm <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ok=j.get("other_kernels") or {}
print(sys.argv[2], "x3 train", j["ms_per_step"], "fwd", (j.get("fwd") or {}).get("ms_per_step"), {k:v.get("avg_launch_us") for k,v in ok.items() if "attention" in k})
PY
done; done
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
