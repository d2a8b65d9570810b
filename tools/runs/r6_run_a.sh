#!/bin/bash
# round 6, call A: (1) tools/vmem_probe (the CU's vector-memory path in B/clk, loads / LDS-DMA / stores, no MFMA) -> profiles/r06_vmem_path_probe.txt;
# (2) stamped traces of the resident kernel at the tile seam (gemm_dev debug bit 8: K-tile 1 | K-tile 2 | rest | realign | epilogue);
# (3) round-6 tests of me_gemm_desc.row_parts; (4) same-box A/B: folded forward with the pairs formed in the consumer GEMM (HEAD) against
# the round-5 route (me_row_stats_combine launches: -DME_NO_ROW_PARTS=1)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6a
mkdir -p $O
cd $R
#timeout 300 tools/_build/vmem_probe > $O/vmem_probe.txt 2>&1; echo "probe rc=$?"
#timeout 300 tools/_build/gemm_dev --iters 20 --check g3:50432:2304:768:0:8 g3:50432:768:768:2:8 g3:50432:3072:768:1:8 g3:50432:768:3072:2:8 > $O/seam_trace.txt 2>&1; echo "trace rc=$?"
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k "partials or refused or combine" > $O/tests_parts.txt 2>&1; echo "tests rc=$?"; tail -5 $O/tests_parts.txt
timeout 600 python -m pytest tests/test_gpu_encoder.py -x -q -k "chains or folded or statistics" > $O/tests_fold.txt 2>&1; echo "tests2 rc=$?"; tail -3 $O/tests_fold.txt
REPS=3 KEEP=$O bash tools/ab_fwd.sh parts=metatransformer_amd/libmetaenc.so combine=tools/_build_prod_noparts/libmetaenc.so > $O/ab_fwd.txt 2>&1
cat $O/ab_fwd.txt
