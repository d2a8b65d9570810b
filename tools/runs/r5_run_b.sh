#!/bin/bash
# round-5 GPU call B: the tile-count planning rule for small NT problems -- every plan checked in full (gemm_dev auto), then the
# reference's shapes end to end (tools/refshapes.py, bf16), round-4 library against this one on the same box, interleaved.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5b
mkdir -p $O
cd $R
G=tools/_build/gemm_dev
for M in 197 1576 3072 4096 6304 8224; do
  CASES=""
  for SH in 2304:768:0 768:768:2 3072:768:1 768:3072:2 768:2304:0 768:3072:0 3072:768:6; do CASES="$CASES auto:$M:$SH"; done
  timeout 300 $G --iters 50 --check $CASES
done 2>&1 | tee $O/small_gemm_auto.txt
cp metatransformer_amd/libmetaenc.so /tmp/cur.so
for rep in 1 2; do
for V in r4 head; do
  [ $V = head ] && cp /tmp/cur.so metatransformer_amd/libmetaenc.so || cp tools/_build_prod_r4/libmetaenc.so metatransformer_amd/libmetaenc.so
  timeout 600 python tools/refshapes.py --quick --dtypes bf16 2>/dev/null | grep bf16 | sed "s/^/$V /"
done
done 2>&1 | tee $O/refshapes_ab.txt
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
for V in r4 head; do
  [ $V = head ] && cp /tmp/cur.so metatransformer_amd/libmetaenc.so || cp tools/_build_prod_r4/libmetaenc.so metatransformer_amd/libmetaenc.so
  timeout 300 python tools/graph_latency.py 2>/dev/null | sed "s/^/$V /"
done 2>&1 | tee $O/latency_ab.txt
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
