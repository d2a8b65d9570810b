#!/bin/bash
# round-4 GPU call AC: whole-problem split-K of small GEMMs when tiles <= slots / 3 (shipped), / 2, / 1 -- the reference's mid-size shapes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4ac
mkdir -p $O
cd $R
cp metatransformer_amd/libmetaenc.so /tmp/cur.so
for rep in 1 2; do
for V in base split2 split1; do
  [ $V = base ] && cp /tmp/cur.so metatransformer_amd/libmetaenc.so || cp tools/_build_prod_$V/libmetaenc.so metatransformer_amd/libmetaenc.so
  for S in xray pointcloud_cls hyperspectral audio; do
    timeout 200 python tools/refshapes.py --only $S --quick 2>/dev/null | grep "bf16" | sed "s/^/$V /"
  done
done
done 2>&1 | tee $O/small_split.txt
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
