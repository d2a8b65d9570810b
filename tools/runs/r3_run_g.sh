#!/bin/bash
# round-3 GPU call G: -m gpu suite + train / fwd kernel traces of the bench (per-kernel averages of the final build)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3g/bench.json').read().strip().splitlines()[-1])
print('train ms',j['ms_per_step'],'value',j['value'],'roof',j['roofline']['frac'],'fwd',j['fwd']['ms_per_step'],j['fwd']['mfma_frac'],j['fwd']['roofline']['frac'])
PY
cd /tmp && export TMPDIR=/tmp
for MODE in fwd train; do
  EXTRA=""; [ $MODE = train ] && EXTRA="--no-fwd-leg"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$MODE -o t -- python $R/bench.py --steps 5 --warmup 2 --mode $MODE --no-cpu-baseline $EXTRA > $O/$MODE.json 2> $O/$MODE.err; echo "prof $MODE rc=$?"
  python $R/tools/prof_summary.py $(find $O/$MODE -name "*kernel_trace.csv" | head -1) 16 > $O/${MODE}_summary.txt; cat $O/${MODE}_summary.txt
done
find $O -name "*agent*" -delete
