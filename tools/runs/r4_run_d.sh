#!/bin/bash
# round-4 GPU call D: the statistics-emitting residual epilogue -- its tests, kernel-level cost (gemm_one-style A/B through
# bench's per-launch records) and the forward with / without the chain; then the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4d
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x -k "row_statistics or chains_layernorm or folded" > $O/pytest_new.log 2>&1; echo "pytest(new) rc=$?"; tail -6 $O/pytest_new.log; grep -E "chained vs|folded vs" $O/pytest_new.log
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
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r4d/bench.json').read().strip().splitlines()[-1])
print('train ms',j['ms_per_step'],'roof',j['roofline']['frac'],'fwd',j['fwd']['ms_per_step'],j['fwd']['mfma_frac'],j['fwd']['roofline']['frac'])
PY
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
