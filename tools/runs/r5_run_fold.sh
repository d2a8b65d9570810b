#!/bin/bash
# round 5: the split-K fold of the Large weight gradients (16 MiB slabs alias in HBM): slab stride padding A/B.
# arms: tools/_build_prod_pad<K>/libmetaenc.so built with -DG3_TN_SLAB_PAD=<floats>  (python -m metatransformer_amd.build --variant padK -DG3_TN_SLAB_PAD=K)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5fold; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for arm in "$@"; do
  lib=$R/metatransformer_amd/libmetaenc.so
  [ "$arm" != "tree" ] && lib=$R/tools/_build_prod_$arm/libmetaenc.so
  rm -rf $O/t
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python $R/tools/fold_time.py --lib $lib > $O/out_$arm.txt 2>&1
  f=$(find $O/t -name "*kernel_trace.csv" | head -1)
  echo "== $arm"
  python - "$f" <<'PY'
import csv, sys, collections
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
agg=collections.OrderedDict()
for r in rows:
    k=r["Kernel_Name"]
    if "splitk_fold" not in k and "g3tn" not in k: continue
    key=("fold" if "fold" in k else "wgrad", r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size",""), r.get("Grid_Size_Y",""))
    d=agg.setdefault(key,[0,0]); d[0]+=1; d[1]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k,(n,t) in agg.items(): print("   %-6s grid %8s x %-6s n=%3d avg %8.1f us" % (k[0],k[1],k[2],n,t/n/1e3))
PY
done
rm -rf $O/t
