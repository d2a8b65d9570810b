#!/bin/bash
# round 6, call P: per-kernel times of the streaming attention backward (rocprofv3 --kernel-trace --stats over tools/attn_time.py), in-tree library
# vs the arms named in $ARMS (tools/_build_prod_<arm>/libmetaenc.so), at N = 1568 (config 5) and 592 (config 4)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6p
mkdir -p $O
cd $R
export TMPDIR=/tmp
for arm in cur ${ARMS:-rows16}; do
  LIB=""; [ $arm != cur ] && LIB="--lib tools/_build_prod_$arm/libmetaenc.so"
  for S in "32 1568 16 64" "128 592 12 64"; do
    tag=${arm}_$(echo $S | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$tag -o t -- python tools/attn_time.py $LIB $S > $O/$tag.txt 2>&1
    f=$(find $O/t_$tag -name '*kernel_stats.csv' | head -1)
    echo "== $tag"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "attn_" in r["Name"]:
        print(f'{r["Name"].split("::")[-1].split("(")[0]:44s} n={r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
    rm -rf $O/t_$tag
  done
done
