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
    cp $P metatransformer_amd/libmetaenc.so
    timeout 400 python bench.py $ARGS > /tmp/ab_$L.json 2> /tmp/ab_$L.err || { echo "$L: bench failed"; tail -3 /tmp/ab_$L.err; continue; }
    python - "$L" <<'PY'
import json,sys
L=sys.argv[1]
j=json.loads(open(f'/tmp/ab_{L}.json').read().strip().splitlines()[-1])
f=j.get('fwd') or {}
print(f"{L:10s} train {j['ms_per_step']:7.3f} ms  gemm roof {j['roofline']['frac']:.4f}  fwd {f.get('ms_per_step',0):6.3f} ms  fwd mfma {f.get('mfma_frac',0):.4f}  fwd gemm roof {(f.get('roofline') or {}).get('frac',0):.4f}")
PY
  done
done
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
