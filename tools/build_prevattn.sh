#!/bin/bash
# Builds tools/_build_base/libmetaenc_prevattn.so: today's library with attention.hip taken from an older commit (default: the last one
# before the ring kernels), for the same-box A/B scripts tools/ab_attn.sh / tools/ab_attn_shapes.sh.   tools/build_prevattn.sh [commit]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=${1:-5e02f14}
T=$(mktemp -d)
git -C $R show $C:metatransformer_amd/csrc/attention.hip > $T/attention.hip
python -m metatransformer_amd.build > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=fast -w -c $T/attention.hip \
    -I $R/metatransformer_amd/csrc -I $R/include -o $T/attention.o
mkdir -p $R/tools/_build_base
OBJS=$(ls $R/metatransformer_amd/csrc/_obj/*.o | grep -v "/attention.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/_build_base/libmetaenc_prevattn.so $OBJS $T/attention.o
echo $R/tools/_build_base/libmetaenc_prevattn.so
