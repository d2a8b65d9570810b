#!/bin/bash
# round-5 GPU call N: 3xbf16 train step, weight gradients on the side stream (default) vs serial order, same box, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5n
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for A in 1 0; do
    ME_WGRAD_OVERLAP=$A timeout 300 python bench.py --dtype fp32 --fp32-mode 3xbf16 --steps 4 --warmup 1 --no-cpu-baseline --no-fwd-leg 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side-stream' if '$A' == '1' else 'serial     ', j['ms_per_step'], j['value'])"
  done
done 2>&1 | tee $O/x3_wgrad_side_ab.txt
