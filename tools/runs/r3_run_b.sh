#!/bin/bash
# round-3 GPU call B: full -m gpu suite, the default bench line (with the CPU baseline leg), forward + train kernel traces
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
/usr/bin/time -v timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json; grep -E "Elapsed|Maximum resident" $O/bench.err
cd /tmp && export TMPDIR=/tmp
for MODE in fwd train; do
  EXTRA=""; [ $MODE = train ] && EXTRA="--no-fwd-leg"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$MODE -o t -- python $R/bench.py --steps 5 --warmup 2 --mode $MODE --no-cpu-baseline $EXTRA > $O/$MODE.json 2> $O/$MODE.err; echo "prof $MODE rc=$?"
  python $R/tools/prof_summary.py $(find $O/$MODE -name "*kernel_trace.csv" | head -1) 30 > $O/${MODE}_summary.txt; cat $O/${MODE}_summary.txt
done
find $O -name "*agent*" -delete
find $O -name "*kernel_trace.csv" -size +20M -delete
