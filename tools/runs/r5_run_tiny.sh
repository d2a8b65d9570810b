#!/bin/bash
# round 5: one-wave-per-head attention for N <= 64 (attention_tiny.hip) against the tiled kernels it replaces there (library built with
# -DME_TINY_ATTN=0: python -m metatransformer_amd.build --variant notiny -DME_TINY_ATTN=0), batches large enough that the launches are not host-bound
mkdir -p gpurun_out/r5tiny
O=gpurun_out/r5tiny/attn_tiny_ab.txt
: > $O
for s in "1024 16 12 64" "2048 8 12 64" "512 32 12 64" "256 48 12 64" "256 64 12 64" "512 50 32 24" "1024 24 32 24" "2048 16 32 24" "512 64 32 24" "256 40 16 40"; do
  for lib in tools/_build_prod_notiny/libmetaenc.so metatransformer_amd/libmetaenc.so; do
    echo "== $lib" >> $O
    python tools/attn_time.py --lib $lib $s 2>&1 | grep -v "amdgpu.ids\|no lse" >> $O
  done
done
for s in "1024 16 12 64" "512 32 12 64" "256 64 12 64" "512 50 32 24" "256 40 16 40"; do
  for lib in tools/_build_prod_notiny/libmetaenc.so metatransformer_amd/libmetaenc.so; do
    echo "== $lib" >> $O
    python tools/attn_time.py --fp32 --lib $lib $s 2>&1 | grep -v "amdgpu.ids\|no lse" >> $O
  done
done
python tools/attn_tiny_fold.py $O
