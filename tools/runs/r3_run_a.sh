#!/bin/bash
# round-3 GPU call A: full -m gpu suite, gemm_dev A/B (round-2 build vs this build), bench line, forward kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
CASES="g3:50432:2304:768:0 g3:50432:3072:768:1 g3:50432:3072:768:7 g3:50432:768:3072:3 g3:50432:768:3072:2 g3:50432:768:768:2 g3:50432:768:3072:6"
for B in _build_base _build; do
  timeout 300 tools/$B/gemm_dev --iters 30 --check $CASES > $O/gemm$B.txt 2>&1; echo "gemm_dev $B rc=$?"
  cat $O/gemm$B.txt
done
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd -o t -- python $R/bench.py --steps 5 --warmup 2 --mode fwd --no-cpu-baseline > $O/fwd.json 2> $O/fwd.err; echo "prof rc=$?"
find $O -name "*agent*" -delete
python $R/tools/prof_summary.py $(find $O/fwd -name "*kernel_trace.csv" | head -1) 25 > $O/fwd_summary.txt; cat $O/fwd_summary.txt
find $O/fwd -name "*kernel_trace.csv" -size +20M -delete
