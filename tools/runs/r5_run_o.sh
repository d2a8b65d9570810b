#!/bin/bash
# round-5 GPU call O: rocprofv3 kernel summaries of the fp32 3xbf16 bench (forward and train)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5o
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd -o t -- python $R/bench.py --dtype fp32 --fp32-mode 3xbf16 --mode fwd --steps 4 --warmup 1 --no-cpu-baseline > $O/fwd.json 2> $O/fwd.err
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python $R/bench.py --dtype fp32 --fp32-mode 3xbf16 --steps 3 --warmup 1 --no-cpu-baseline --no-fwd-leg > $O/train.json 2> $O/train.err
find $O -name "*agent*" -delete
cd $R
python tools/prof_summary.py $(find $O/fwd -name "*kernel_trace.csv" | head -1) 14 > $O/fwd_summary.txt
python tools/prof_summary.py $(find $O/train -name "*kernel_trace.csv" | head -1) 24 > $O/train_summary.txt
head -20 $O/fwd_summary.txt
