#!/bin/bash
# round-4 GPU call F: kernel-level cost of the statistics epilogue and of 128-row items in the residual kernel (sustained loops);
# per-kernel profile of the small-batch forward, round-1 snapshot against HEAD
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4f
mkdir -p $O
cd $R
CASES="g3:50432:768:768:2 g3:50432:768:768:8 g3:50432:768:3072:2 g3:50432:768:3072:8"
for round in 1 2; do
  echo "== base (pass $round)"; timeout 300 tools/_build/gemm_dev --check --iters 30 --power 0.8 $CASES 2>&1 | tee $O/gd_base_$round.txt | grep -E "TF/s|power:"
  echo "== hi2 (pass $round)"; timeout 300 tools/_build_hi2/gemm_dev --check --iters 30 --power 0.8 g3:50432:768:768:2 g3:50432:768:3072:2 2>&1 | tee $O/gd_hi2_$round.txt | grep -E "TF/s|power:"
done
cd /tmp && export TMPDIR=/tmp
for P in r1 head; do
  PK=""; [ $P = r1 ] && PK="--pkg $R/tools/_build_r1"
  for B in 1 8; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lat_${P}_$B -o t -- python $R/tools/graph_latency.py $PK --batches $B > $O/lat_${P}_$B.log 2>&1
    echo "== $P B=$B"; grep "B=" $O/lat_${P}_$B.log
    python $R/tools/prof_summary.py $(find $O/lat_${P}_$B -name "*kernel_trace.csv" | head -1) 14 2>/dev/null | tee $O/lat_${P}_${B}_summary.txt | head -24
  done
done
find $O -name "*agent*" -delete; find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
