#!/bin/bash
# round 6, call C: socket power / clock over sustained forward and train loops, with idle "rests" behind every Block (tools/power_probe.py)
# -> profiles/r06_power.txt; ADVICE-r5 regression tests
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6c
mkdir -p $O
cd $R
timeout 300 python tools/power_probe.py --mode fwd --secs 3 --rest-us 0 10 20 50 100 0 > $O/power_fwd.txt 2>&1; echo "fwd rc=$?"; cat $O/power_fwd.txt
timeout 300 python tools/power_probe.py --mode train --secs 3 --rest-us 0 50 0 > $O/power_train.txt 2>&1; echo "train rc=$?"; cat $O/power_train.txt
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q -k "optimizer or pack_encoder" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
