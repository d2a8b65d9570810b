#!/bin/bash
# round-4 GPU call T: forward with the LayerNorm statistics chained out of the residual epilogues against the stand-alone statistics
# pass (bench.py --no-chain-stats), three interleaved pairs; train step of the same build
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4t
mkdir -p $O
cd $R
for rep in 1 2 3; do
for A in "chain" "nochain --no-chain-stats"; do
  set -- $A
  timeout 300 python bench.py --steps 10 --warmup 3 --mode fwd --no-cpu-baseline $2 > $O/fwd_$1_$rep.json 2> $O/fwd_$1_$rep.err || { echo "$1 failed"; tail -3 $O/fwd_$1_$rep.err; }
  python - $O/fwd_$1_$rep.json $1 <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ok=j.get('other_kernels') or {}
us=lambda k: (ok.get(k) or {}).get('avg_launch_us',0)
n=lambda k: (ok.get(k) or {}).get('launches_per_step',0)
print(f"{sys.argv[2]:8s} fwd {j['ms_per_step']:6.3f} ms  mfma {j['mfma_frac_end_to_end']:.4f}  gemm {j['roofline']['avg_launch_us']:6.1f} us x{j['roofline']['launches_per_step']} roof {j['roofline']['frac']:.4f}  stats {us('row_stats'):5.1f} us x{n('row_stats')}  attn {us('attention_fwd'):5.1f}")
PY
done
done 2>&1 | tee $O/chain_ab.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fwd-leg 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', j['ms_per_step'], 'roof', j['roofline']['frac'])" | tee $O/train.txt
