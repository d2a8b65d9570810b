#!/bin/bash
# round-5 GPU call P: rocprofv3 kernel summary of the reference's small batches (frozen encoder fwd + dx), bf16
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for S in timeseries xray; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$S -o t -- python $R/tools/refshapes.py --only $S --quick --dtypes bf16 > $O/$S.txt 2> $O/$S.err
done
find $O -name "*agent*" -delete
cd $R
for S in timeseries xray; do echo "== $S"; python tools/prof_summary.py $(find $O/$S -name "*kernel_trace.csv" | head -1) 26; done
