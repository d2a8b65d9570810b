#!/bin/bash
# Compile-time A/B arms of the cache policies (gemm3_core.h G3_POL_*, common.h ME_POL_*: 2 = nt, 16 = sc1).
#   dev arms     -> tools/_build_<arm>/      (libmetaenc_dev.so + gemm_dev, kernel-level A/B with power / clock read-outs)
#   product arms -> tools/_build_prod_<arm>/ (libmetaenc.so, swapped in by tools/ab_bench.sh for bench-level same-box A/B)
# All git-ignored; they travel to the GPU box with the snapshot.
set -e
cd "$(dirname "$0")/.."
B="python -m metatransformer_amd.build"
if [ "$1" = dev ]; then
  $B --dev > /dev/null
  $B --dev --variant cnt      -DG3_POL_C=2  -DG3_POL_P=2 > /dev/null
  $B --dev --variant csc1     -DG3_POL_C=16 -DG3_POL_P=16 > /dev/null
  $B --dev --variant csc1rnt  -DG3_POL_C=16 -DG3_POL_P=16 -DG3_POL_R=2 > /dev/null
  $B --dev --variant csc1ant  -DG3_POL_C=16 -DG3_POL_P=16 -DG3_POL_R=2 -DG3_POL_A=2 > /dev/null
  $B --dev --variant cntant   -DG3_POL_C=2  -DG3_POL_P=2  -DG3_POL_R=2 -DG3_POL_A=2 > /dev/null
else
  NT="-DG3_POL_C=2 -DG3_POL_P=2"
  $B --variant cnt     $NT > /dev/null
  $B --variant cntr    $NT -DG3_POL_R=2 > /dev/null
  $B --variant cntattn $NT -DME_POL_ATTN_ST=2 > /dev/null
  $B --variant cntlnst $NT -DME_POL_LN_ST=1 > /dev/null
  $B --variant cntlnld $NT -DME_POL_LN_LD=1 > /dev/null
  $B --variant cntslab $NT -DME_POL_SLAB=1 > /dev/null
fi
ls -d tools/_build*/
