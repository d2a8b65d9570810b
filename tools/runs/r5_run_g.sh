#!/bin/bash
# round-5 GPU call G: x3 attention forward (tests + bench), the fixed attention backward check
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5g
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -q -k "x3 or 3xbf16 or split3 or three_plane" 2>&1 | tail -25 | tee $O/pytest_x3.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention_bwd" 2>&1 | tail -8 | tee $O/pytest_attn_bwd.txt
timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q -k "fold or inference or chain" 2>&1 | tail -8 | tee $O/pytest_fold.txt
timeout 600 python bench.py --dtype fp32 --fp32-mode 3xbf16 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_fp32_3xbf16.json 2> $O/bench_fp32_3xbf16.err || tail -5 $O/bench_fp32_3xbf16.err
python -c "import json; j=json.load(open('$O/bench_fp32_3xbf16.json')); print('3xbf16 train', j['ms_per_step'], 'fwd', j['fwd']['ms_per_step'], j['fwd']['mfma_frac'], j['fwd']['other_kernels'])" 2>&1 | tee $O/bench.txt
