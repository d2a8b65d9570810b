#!/bin/bash
# same-box A/B of two library builds on the attention entry points: tools/ab_libs.sh A.so B.so "B N H hd" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
LA=$1; LB=$2; shift 2
cp metatransformer_amd/libmetaenc.so /tmp/cur.so
for rep in 1 2; do
for V in $LA $LB; do
  cp $V metatransformer_amd/libmetaenc.so
  for S in "$@"; do echo -n "$(basename $V) "; python tools/attn_time.py $S 2>&1 | grep -E "^fwd  |^bwd" | sed "s/            / /g" | tr '\n' ' '; echo; done
done; done
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
