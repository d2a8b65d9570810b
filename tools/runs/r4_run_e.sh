#!/bin/bash
# round-4 GPU call E: chain / no-chain forward A/B, fp32 bench line, reference-shape sweep, small-batch latency A/B against the
# round-1 snapshot, whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4e
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
done
timeout 600 python bench.py --dtype fp32 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err; echo "fp32 bench rc=$?"; tail -2 $O/bench_fp32.err
python - <<'PY'
import json
try:
    j=json.loads(open('gpurun_out/r4e/bench_fp32.json').read().strip().splitlines()[-1])
    print('fp32 train ms',j['ms_per_step'],'roof',j['roofline']['achieved'],j['roofline']['frac'],'fwd',j['fwd']['ms_per_step'],j['fwd']['mfma_frac'],j['fwd']['roofline']['achieved'])
except Exception as e: print('fp32 parse failed',e)
PY
timeout 900 python tools/refshapes.py --out $O/refshapes.json 2>&1 | tee $O/refshapes.txt | tail -20
for rep in 1 2; do
  timeout 300 python tools/graph_latency.py --pkg tools/_build_r1 2>&1 | grep -E "package|B=" | tee -a $O/latency_r1.txt
  timeout 300 python tools/graph_latency.py 2>&1 | grep -E "package|B=" | tee -a $O/latency_head.txt
done
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
