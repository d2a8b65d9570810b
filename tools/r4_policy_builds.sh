#!/bin/bash
# Compile-time A/B arms of the resident NT GEMM's cache policies (gemm3_core.h, G3_POL_*: 2 = nt, 16 = sc1): one dev library +
# gemm_dev driver per arm under tools/_build_<arm>/ (git-ignored; they travel to the GPU box with the snapshot).
set -e
cd "$(dirname "$0")/.."
python -m metatransformer_amd.build --dev > /dev/null
python -m metatransformer_amd.build --dev --variant cnt      -DG3_POL_C=2  -DG3_POL_P=2 > /dev/null
python -m metatransformer_amd.build --dev --variant csc1     -DG3_POL_C=16 -DG3_POL_P=16 > /dev/null
python -m metatransformer_amd.build --dev --variant csc1rnt  -DG3_POL_C=16 -DG3_POL_P=16 -DG3_POL_R=2 > /dev/null
python -m metatransformer_amd.build --dev --variant csc1ant  -DG3_POL_C=16 -DG3_POL_P=16 -DG3_POL_R=2 -DG3_POL_A=2 > /dev/null
python -m metatransformer_amd.build --dev --variant cntant   -DG3_POL_C=2  -DG3_POL_P=2  -DG3_POL_R=2 -DG3_POL_A=2 > /dev/null
ls -d tools/_build*/
