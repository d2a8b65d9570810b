#!/bin/bash
# round 6, call W: GELU and its derivative from ONE Phi / Gaussian in the epilogues that save gelu' (EPI 9 of the three-product mode, EPI 7 of the
# one-tile kernels, the generic epilogue): GEMM / MLP tests, fp32 bench lines and the reference's batch shapes against the commit before
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6w
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "gelu or saved or mlp or 3xbf16 or three or two_plane or x3" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
cp metatransformer_amd/libmetaenc.so /tmp/cur.so
for rep in 1 2; do for arm in cur base; do
  [ $arm = base ] && cp tools/_build_prod_pairbase/libmetaenc.so metatransformer_amd/libmetaenc.so || cp /tmp/cur.so metatransformer_amd/libmetaenc.so
  timeout 600 python bench.py --dtype fp32 --fp32-mode 3xbf16 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_${arm}_$rep.json 2> $O/bench.err
  python - "$O/bench_${arm}_$rep.json" $arm <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "x3 train", j["ms_per_step"], "fwd", (j.get("fwd") or {}).get("ms_per_step"))
PY
done; done
for arm in cur base; do
  [ $arm = base ] && cp tools/_build_prod_pairbase/libmetaenc.so metatransformer_amd/libmetaenc.so || cp /tmp/cur.so metatransformer_amd/libmetaenc.so
  echo "== refshapes $arm"; timeout 600 python tools/refshapes.py --dtypes fp32x3,bf16 --only timeseries 2>&1 | grep -E "fp32x3|bf16"; timeout 600 python tools/refshapes.py --dtypes fp32x3,bf16 --only pointcloud_cls 2>&1 | grep -E "fp32x3|bf16"
done
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
