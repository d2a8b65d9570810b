#!/bin/bash
# round-4 GPU call Z: priority of me_block_bwd's side stream (default / high / low) and the serial order, bench interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4z
mkdir -p $O
cd $R
for rep in 1 2; do
for A in "default" "high" "low" "serial"; do
  if [ $A = serial ]; then export ME_WGRAD_OVERLAP=0; unset ME_WGRAD_PRIO; else export ME_WGRAD_OVERLAP=1; export ME_WGRAD_PRIO=$A; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fwd-leg 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$A', j['ms_per_step'], j['value'], j['schedule']['serial_ms_per_step'])"
done
done 2>&1 | tee $O/prio.txt
