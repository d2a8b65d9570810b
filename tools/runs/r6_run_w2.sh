#!/bin/bash
# round 6, call W2: re-check of the small-batch bf16 lines of call W (order swapped, three repetitions)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp metatransformer_amd/libmetaenc.so /tmp/cur.so
for rep in 1 2 3; do for arm in base cur; do
  [ $arm = base ] && cp tools/_build_prod_pairbase/libmetaenc.so metatransformer_amd/libmetaenc.so || cp /tmp/cur.so metatransformer_amd/libmetaenc.so
  echo -n "$arm: "; timeout 600 python tools/refshapes.py --dtypes bf16 --only timeseries 2>&1 | grep -E "bf16" | cut -c1-110
done; done
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
