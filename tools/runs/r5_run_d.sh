#!/bin/bash
# round-5 GPU call D: round-5 tests (small shapes, inference mode, graph replay, 3xbf16) + the fp32 bench in both fp32 modes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5d
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -q -x 2>&1 | tail -30 | tee $O/pytest_r5.txt
for MODE in 3xbf16 exact; do
  timeout 600 python bench.py --dtype fp32 --fp32-mode $MODE --mode fwd --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_fp32_${MODE}_fwd.json 2> $O/bench_fp32_${MODE}_fwd.err || tail -5 $O/bench_fp32_${MODE}_fwd.err
  python -c "import json; j=json.load(open('$O/bench_fp32_${MODE}_fwd.json')); print('$MODE fwd', j['ms_per_step'], 'ms', j['value'], j['roofline']['achieved'], j['roofline']['frac'], {k:v['avg_launch_us'] for k,v in j['other_kernels'].items()})"
done 2>&1 | tee $O/bench_fp32.txt
timeout 900 python bench.py --dtype fp32 --fp32-mode 3xbf16 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_fp32_3xbf16_train.json 2> $O/bench_fp32_3xbf16_train.err || tail -5 $O/bench_fp32_3xbf16_train.err
python -c "import json; j=json.load(open('$O/bench_fp32_3xbf16_train.json')); print('3xbf16 train', j['ms_per_step'], 'ms', j['value'], j['roofline'], j['other_kernels'])" 2>&1 | tee -a $O/bench_fp32.txt
