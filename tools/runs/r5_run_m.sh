#!/bin/bash
# round-5 GPU call M: overlapped optimizer (per-Block AdamW + zero-fill + transposes on a side stream): test + same-box A/B of the train step
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5m
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_optim.py -m gpu -q 2>&1 | tail -12 | tee $O/pytest_optim.txt
for rep in 1 2 3; do
  for A in "" "--no-opt-overlap"; do
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fwd-leg $A 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap' if '$A' == '' else 'one-pass', j['ms_per_step'], j['value'], j['schedule']['serial_ms_per_step'])"
  done
done 2>&1 | tee $O/opt_overlap_ab.txt
