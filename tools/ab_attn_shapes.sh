#!/bin/bash
# same-box A/B of the attention entry points alone over a list of shapes: current library vs $PREV (default: tools/_build_base/libmetaenc_prevattn.so,
# built by tools/build_prevattn.sh)
PREV=${PREV:-tools/_build_base/libmetaenc_prevattn.so}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp metatransformer_amd/libmetaenc.so /tmp/cur.so
for V in cur prev; do
  [ $V = cur ] && cp /tmp/cur.so metatransformer_amd/libmetaenc.so || cp $PREV metatransformer_amd/libmetaenc.so
  for S in "$@"; do echo -n "$V "; python tools/attn_time.py $S 2>&1 | grep -E "^fwd  |^bwd" | tr '\n' ' '; echo; done
done
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
