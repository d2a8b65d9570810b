#!/bin/bash
# same-box A/B of library builds on the FORWARD bench (bench.py --mode fwd): tools/ab_fwd.sh "label=path.so" ... ; REPS interleaved runs per arm.
# The in-tree library is restored at the end.  (tools/ab_bench.sh is the train-step form.)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
REPS=${REPS:-3}
ARGS=${ARGS:---steps 20 --warmup 5 --no-cpu-baseline --mode fwd}
cp metatransformer_amd/libmetaenc.so /tmp/cur.so
for rep in $(seq $REPS); do
  for arm in "$@"; do
    L=${arm%%=*}; P=${arm#*=}
    [ "$P" = "metatransformer_amd/libmetaenc.so" ] && P=/tmp/cur.so      # (the in-tree library itself as an arm: its saved copy)
    cp $P metatransformer_amd/libmetaenc.so
    timeout 400 python bench.py $ARGS > /tmp/abf_$L.json 2> /tmp/abf_$L.err || { echo "$L: bench failed"; tail -3 /tmp/abf_$L.err; continue; }
    [ -n "$KEEP" ] && cp /tmp/abf_$L.json $KEEP/abf_${L}_$rep.json
    python - "$L" <<'PY'
import json,sys
L=sys.argv[1]
j=json.loads(open(f'/tmp/abf_{L}.json').read().strip().splitlines()[-1])
ok=j.get('other_kernels') or {}
us=lambda k: (ok.get(k) or {}).get('avg_launch_us',0)
n=lambda k: (ok.get(k) or {}).get('launches_per_step',0)
r=j['roofline']
print(f"{L:10s} fwd {j['ms_per_step']:7.3f} ms  mfma_end_to_end {j['mfma_frac_end_to_end']:.4f}  gemm {r['avg_launch_us']:6.1f} us x{r['launches_per_step']} roof {r['frac']:.4f}  row_stats {us('row_stats'):5.1f} us x{n('row_stats')}  attn {us('attention_fwd'):5.1f}")
PY
  done
done
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
