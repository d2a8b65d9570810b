#!/bin/bash
# round 6, call I: kernel traces of the contention rehearsal (R = 16 held CUs), weight gradients planned with and without the reservation:
# which launches pay, and how much (tools/contention_trace.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for RES in 0 16; do
  rm -rf $O/t$RES
  ME_WGRAD_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t$RES -o t -- python $R/tools/contention.py --cus 16 --reserve $RES --steps 3 > $O/run_$RES.log 2>&1
  f=$(find $O/t$RES -name "*kernel_trace.csv" | head -1)
  echo "== serial order, reserve $RES"; python $R/tools/contention_trace.py $f | tee $O/trace_serial_res$RES.txt
  rm -rf $O/t$RES
done
