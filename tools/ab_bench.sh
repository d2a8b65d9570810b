#!/bin/bash
# same-box A/B of library builds at the bench level: tools/ab_bench.sh "label=path.so" ... ; each arm runs bench.py (train + fwd leg)
# REPS times, interleaved; prints train ms / fwd ms per arm and run.  The in-tree library is restored at the end.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
REPS=${REPS:-2}
ARGS=${ARGS:---steps 10 --warmup 3 --no-cpu-baseline}
cp metatransformer_amd/libmetaenc.so /tmp/cur.so
for rep in $(seq $REPS); do
  for arm in "$@"; do
    L=${arm%%=*}; P=${arm#*=}
    [ "$P" = "metatransformer_amd/libmetaenc.so" ] && P=/tmp/cur.so      # (the in-tree library itself as an arm: its saved copy)
    cp $P metatransformer_amd/libmetaenc.so
    timeout 400 python bench.py $ARGS > /tmp/ab_$L.json 2> /tmp/ab_$L.err || { echo "$L: bench failed"; tail -3 /tmp/ab_$L.err; continue; }
    [ -n "$KEEP" ] && cp /tmp/ab_$L.json $KEEP/ab_${L}_$rep.json
    python - "$L" <<'PY'
import json,sys
L=sys.argv[1]
j=json.loads(open(f'/tmp/ab_{L}.json').read().strip().splitlines()[-1])
f=j.get('fwd') or {}
ok=j.get('other_kernels') or {}
fo=f.get('other_kernels') or {}
us=lambda d,k: (d.get(k) or {}).get('avg_launch_us',0)
print(f"{L:10s} train {j['ms_per_step']:7.3f} ms  gemm {j['roofline']['avg_launch_us']:6.1f} us roof {j['roofline']['frac']:.4f} wgrad {j['roofline']['wgrad_kernel']['avg_launch_us']:6.1f}  ln f/b {us(ok,'layernorm_fwd'):5.1f}/{us(ok,'layernorm_bwd'):5.1f}  attn f/b {us(ok,'attention_fwd'):5.1f}/{us(ok,'attention_bwd'):6.1f} | fwd {f.get('ms_per_step',0):6.3f} ms mfma {f.get('mfma_frac',0):.4f} gemm {(f.get('roofline') or {}).get('avg_launch_us',0):6.1f} us  stats {us(fo,'row_stats'):5.1f} attn {us(fo,'attention_fwd'):5.1f}")
PY
  done
done
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
