#!/bin/bash
# round-4 GPU call U: whole GPU suite on HEAD, then the round's profile set (tools/prof_round.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4u
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/prof_round.sh
