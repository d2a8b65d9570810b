#!/bin/bash
# round-5 GPU call E: the remaining round-5 tests (3xbf16, inference mode, graph replay), the CU-hog test, the contention table
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -q -k "not small_m and not reference_batch" 2>&1 | tail -30 | tee $O/pytest_r5.txt
timeout 600 python -m pytest tests/test_gpu_comm.py -m gpu -q -k "holds_cus" 2>&1 | tail -15 | tee $O/pytest_hog.txt
timeout 900 python tools/contention.py --cus 0,8,16,32,64 --out $O/contention.txt 2>&1 | tail -12
python __graft_entry__.py smoke 2>&1 | tail -6 | tee $O/smoke.txt
