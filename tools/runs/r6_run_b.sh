#!/bin/bash
# round 6, call B: round-6 tests in full (row_parts + the full-size 3xbf16 parity tests, slow ones included), the statistics / folded tests of
# round 4, and the ADVICE-r5 regression tests
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6b
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_gpu_round6.py -x -q -s > $O/tests_round6.txt 2>&1; echo "round6 rc=$?"; tail -8 $O/tests_round6.txt
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -k "chains or folded or statistics or inference" > $O/tests_fold.txt 2>&1; echo "fold rc=$?"; tail -3 $O/tests_fold.txt
