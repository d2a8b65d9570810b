#!/bin/bash
# round 6, call L: ME_BF16X2 (two-plane MLP tensors, wrapped A operand) in the three-product mode: GEMM tests, every 3xbf16 parity test incl. the
# full-size one, bench lines of config 2 in fp32 (train + forward)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6l
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k "two_plane" > $O/tests_x2.txt 2>&1; echo "x2 rc=$?"; tail -4 $O/tests_x2.txt
timeout 2400 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -x -q -k "3xbf16 or three or x3 or layer_scale or plane" > $O/tests_x3.txt 2>&1; echo "x3 rc=$?"; tail -4 $O/tests_x3.txt
timeout 600 python bench.py --dtype fp32 --fp32-mode 3xbf16 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_fp32_3xbf16.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r6l/bench_fp32_3xbf16.json").read().strip().splitlines()[-1])
print("x3 train", j["ms_per_step"], "fwd", (j.get("fwd") or {}).get("ms_per_step"), j.get("power"))
PY
