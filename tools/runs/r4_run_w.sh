#!/bin/bash
# round-4 GPU call W: weight-gradient GEMMs on the side stream of me_block_bwd (ME_WGRAD_OVERLAP=1, the default) against the serial
# order (=0): backward parity / optimizer / communication tests with the overlap on, then the bench interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4w
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_optim.py tests/test_gpu_comm.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for rep in 1 2 3; do
for A in 1 0; do
  ME_WGRAD_OVERLAP=$A timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fwd-leg > $O/train_$A_$rep.json 2> $O/train_$A_$rep.err || { echo "overlap=$A failed"; tail -3 $O/train_$A_$rep.err; }
  python - $O/train_$A_$rep.json $A <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ok=j.get('other_kernels') or {}
us=lambda k: (ok.get(k) or {}).get('avg_launch_us',0)
print(f"overlap={sys.argv[2]} train {j['ms_per_step']:7.3f} ms  gemm {j['roofline']['avg_launch_us']:6.1f} us roof {j['roofline']['frac']:.4f} wgrad {j['roofline']['wgrad_kernel']['avg_launch_us']:6.1f}  ln f/b {us('layernorm_fwd'):5.1f}/{us('layernorm_bwd'):5.1f}  attn f/b {us('attention_fwd'):5.1f}/{us('attention_bwd'):6.1f}")
PY
done
done 2>&1 | tee $O/overlap_ab.txt
