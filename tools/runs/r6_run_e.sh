#!/bin/bash
# round 6, call E: weight gradients as balanced static parts (me_gemm_reserve_cus): tests, then the contention rehearsal with and without the
# reservation (tools/contention.py --reserve R / 0) -> profiles/r06_contention.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k "balanced" > $O/tests_sk.txt 2>&1; echo "sk rc=$?"; tail -4 $O/tests_sk.txt
timeout 900 python -m pytest tests/test_gpu_comm.py -x -q -k "cus or hog or reservation" > $O/tests_hog.txt 2>&1; echo "hog rc=$?"; tail -4 $O/tests_hog.txt
timeout 600 python tools/contention.py --cus 0,8,16,32,64 --reserve 0 --out $O/contention_res0.txt > /dev/null 2>&1; echo "c0 rc=$?"; cat $O/contention_res0.txt
timeout 600 python tools/contention.py --cus 0,8,16,32,64 --reserve R --out $O/contention_resR.txt > /dev/null 2>&1; echo "cR rc=$?"; cat $O/contention_resR.txt
timeout 600 python tools/contention.py --cus 0,8,16,32 --reserve 16 --out $O/contention_res16.txt > /dev/null 2>&1; echo "c16 rc=$?"; cat $O/contention_res16.txt
