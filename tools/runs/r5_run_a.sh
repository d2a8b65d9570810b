#!/bin/bash
# round-5 GPU call A: the NT kernels on the reference's small shapes: the shipped plan (auto) against the two-workgroup
# family WITHOUT a K split -- 128 x 256 (g2bw), 128 x 128 (g2bn) and the new 64 x 128 (g2bs) tiles.  Every case checked in full.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5a
mkdir -p $O
cd $R
G=tools/_build/gemm_dev
for M in 197 1576 3072 6304 12864; do
  CASES=""
  for SH in 2304:768:0 768:768:2 3072:768:1 768:3072:2; do
    for F in auto g2bw g2bn g2bs; do CASES="$CASES $F:$M:$SH"; done
  done
  timeout 300 $G --iters 50 --check $CASES
done 2>&1 | tee $O/small_gemm.txt
