#!/bin/bash
# round-4 GPU call N: LayerNorm backward grid sized by the occupancy query; 128-row half items in the statistics epilogue -- full
# GPU suite, then bench A/B against the previous LayerNorm build
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4n
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
P=tools/_build_prod
KEEP=$O REPS=3 bash tools/ab_bench.sh new=/tmp/cur.so prevln=${P}_prevln/libmetaenc.so 2>&1 | tee $O/ab_bench.txt
