#!/bin/bash
# round-4 GPU call AB: the reference's small shapes (bf16, frozen encoder) with the whole-problem split-K of small GEMMs as shipped
# (tiles <= slots / 3), only when tiles <= slots / 6, and never
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4ab
mkdir -p $O
cd $R
cp metatransformer_amd/libmetaenc.so /tmp/cur.so
for rep in 1 2; do
for V in base split6 nosplit; do
  [ $V = base ] && cp /tmp/cur.so metatransformer_amd/libmetaenc.so || cp tools/_build_prod_$V/libmetaenc.so metatransformer_amd/libmetaenc.so
  for S in timeseries tabular graph xray pointcloud_cls; do
    timeout 200 python tools/refshapes.py --only $S --quick 2>/dev/null | grep "bf16" | sed "s/^/$V /"
  done
done
done 2>&1 | tee $O/small_split.txt
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
for V in base nosplit; do
  [ $V = base ] || cp tools/_build_prod_$V/libmetaenc.so metatransformer_amd/libmetaenc.so
  timeout 200 python tools/graph_latency.py --batches 1,8,32 2>&1 | grep "B=" | sed "s/^/$V /"
done 2>&1 | tee $O/latency.txt
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
