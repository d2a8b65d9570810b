#!/bin/bash
# round 6, soak: the whole GPU suite twice more on one box + the attention race hunt with 400 launches per shape (final tree)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6soak
mkdir -p $O
cd $R
for i in 1 2; do
  timeout 2400 python -m pytest tests -m gpu -q --maxfail=5 -p no:cacheprovider > $O/suite_$i.txt 2>&1; echo "suite $i rc=$?"; tail -2 $O/suite_$i.txt
done
timeout 1500 python tools/attn_stress.py 400 > $O/race.txt 2>&1; echo "race rc=$?"; grep launches $O/race.txt
