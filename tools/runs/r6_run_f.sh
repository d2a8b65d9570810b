#!/bin/bash
# round 6, call F: the contention rehearsal in SERIAL order (weight gradients on the main stream: ME_WGRAD_OVERLAP=0), reservation off / on
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6f
mkdir -p $O
cd $R
ME_WGRAD_OVERLAP=0 timeout 600 python tools/contention.py --cus 0,16,32 --reserve 0 --out $O/serial_res0.txt > /dev/null 2>&1; echo "s0 rc=$?"; grep -v "^#" $O/serial_res0.txt
ME_WGRAD_OVERLAP=0 timeout 600 python tools/contention.py --cus 0,16,32 --reserve R --out $O/serial_resR.txt > /dev/null 2>&1; echo "sR rc=$?"; grep -v "^#" $O/serial_resR.txt
