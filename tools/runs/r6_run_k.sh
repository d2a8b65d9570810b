#!/bin/bash
# round 6, call K: resident NT kernel with a claimed FIRST item under me_gemm_reserve_cus (+ the balanced weight-gradient plan): tests, kernel
# traces under a 16-CU hold, step tables
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6k
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k "balanced or claims" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
timeout 900 python -m pytest tests/test_gpu_comm.py -x -q -k "cus or hog or reservation" > $O/tests_hog.txt 2>&1; echo "hog rc=$?"; tail -3 $O/tests_hog.txt
cd /tmp && export TMPDIR=/tmp
for RES in 0 16; do
  rm -rf $O/t$RES
  ME_WGRAD_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t$RES -o t -- python $R/tools/contention.py --cus 16 --reserve $RES --steps 3 > $O/run_$RES.log 2>&1
  f=$(find $O/t$RES -name "*kernel_trace.csv" | head -1)
  echo "== serial order, reserve $RES"; python $R/tools/contention_trace.py $f | tee $O/trace_serial_res$RES.txt
  rm -rf $O/t$RES
done
cd $R
timeout 600 python tools/contention.py --cus 0,8,16,32 --reserve 0 --out $O/contention_res0.txt > /dev/null 2>&1; grep -v "^#" $O/contention_res0.txt
timeout 600 python tools/contention.py --cus 0,8,16,32 --reserve 16 --out $O/contention_res16.txt > /dev/null 2>&1; grep -v "^#" $O/contention_res16.txt
ME_WGRAD_OVERLAP=0 timeout 600 python tools/contention.py --cus 0,16,32 --reserve 0 --out $O/serial_res0.txt > /dev/null 2>&1; grep -v "^#" $O/serial_res0.txt
ME_WGRAD_OVERLAP=0 timeout 600 python tools/contention.py --cus 0,16,32 --reserve 16 --out $O/serial_res16.txt > /dev/null 2>&1; grep -v "^#" $O/serial_res16.txt
