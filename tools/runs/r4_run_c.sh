#!/bin/bash
# round-4 GPU call C: bench-level A/B of the further nt arms on top of the nt output stores; the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c
mkdir -p $O
cd $R
P=tools/_build_prod
KEEP=$O REPS=2 bash tools/ab_bench.sh base=/tmp/cur.so cnt=${P}_cnt/libmetaenc.so cntr=${P}_cntr/libmetaenc.so cntattn=${P}_cntattn/libmetaenc.so cntlnst=${P}_cntlnst/libmetaenc.so cntlnld=${P}_cntlnld/libmetaenc.so cntslab=${P}_cntslab/libmetaenc.so 2>&1 | tee $O/ab_bench.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
