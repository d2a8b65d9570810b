#!/bin/bash
# round-5 GPU call J: the full GPU suite on the final library + the fp32 bench lines (exact / 3xbf16) + PMC of the x3 attention
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5j
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | tail -15 | tee $O/pytest.txt
timeout 600 python bench.py --dtype fp32 --fp32-mode 3xbf16 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_fp32_3xbf16.json 2> $O/bench_fp32_3xbf16.err || tail -5 $O/bench_fp32_3xbf16.err
python -c "import json; j=json.load(open('$O/bench_fp32_3xbf16.json')); print('3xbf16 train', j['ms_per_step'], 'fwd', j['fwd']['ms_per_step'], j['fwd']['mfma_frac'], {k: v['avg_launch_us'] for k, v in j['fwd']['other_kernels'].items()}, {k: v['avg_launch_us'] for k, v in j['other_kernels'].items()})" 2>&1 | tee $O/bench.txt
python __graft_entry__.py smoke 2>&1 | tail -5 | tee $O/smoke.txt
