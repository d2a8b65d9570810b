#!/bin/bash
# round 6: rocprofv3 kernel summary of the fp32-as-3xbf16 train step (two-plane MLP tensors) -> profiles/r06_fp32_3xbf16_train_summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6x3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python $R/bench.py --dtype fp32 --fp32-mode 3xbf16 --steps 4 --warmup 2 --no-cpu-baseline --no-fwd-leg > $O/train.json 2> $O/train.err
echo "rc=$?"
find $O -name "*agent*" -delete
f=$(find $O/train -name "*kernel_stats.csv" | head -1)
python - "$f" > $O/fp32_3xbf16_train_summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# rocprofv3 --kernel-trace --stats: python bench.py --dtype fp32 --fp32-mode 3xbf16 --steps 4 --warmup 2  (6 steps under the profiler)")
print(f"{'kernel':90s} {'calls':>7s} {'avg us':>9s} {'share':>7s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:9.1f} {100*float(r['TotalDurationNs'])/tot:6.2f}%")
PY
cat $O/fp32_3xbf16_train_summary.txt | cut -c1-130
tail -1 $O/train.json | cut -c1-300
rm -rf $O/train
