#!/bin/bash
# round-5 GPU call Q: rocprofv3 kernel summaries of the Large configurations (config 3 [128,512,1024], config 5 shape [32,1568,1024]) and the mixed one
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5q
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in large512 large1568 mixed; do
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$W -o t -- python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-fwd-leg > $O/$W.json 2> $O/$W.err
done
find $O -name "*agent*" -delete
cd $R
for W in large512 large1568 mixed; do echo "== $W"; python tools/prof_summary.py $(find $O/$W -name "*kernel_trace.csv" | head -1) 22 | tee $O/${W}_summary.txt; done
