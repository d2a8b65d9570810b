#!/bin/bash
# round 6, call U: where the time goes at the reference's own batch sizes in the default fp32 arithmetic (three-product mode): kernel summaries of
# tools/refshapes.py --dtypes fp32x3 for two recipes (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6u
mkdir -p $O
cd $R
export TMPDIR=/tmp
for name in pointcloud_cls timeseries_forecast; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$name -o t -- python tools/refshapes.py --only $name --dtypes fp32x3 > $O/$name.txt 2>&1
  cat $O/$name.txt | grep fp32x3
  T=$(find $O/t_$name -name "*kernel_trace.csv" | head -1)
  python tools/prof_summary.py $T 22 > $O/${name}_summary.txt; cut -c1-150 $O/${name}_summary.txt
  rm -rf $O/t_$name
done
