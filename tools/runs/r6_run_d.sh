#!/bin/bash
# round 6, call D: the whole GPU suite on the tree with Block.default_fp32_mode = "3xbf16" (which tests pinned the exact path implicitly?)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6d
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $O/gpu_suite.txt 2>&1; echo "suite rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" $O/gpu_suite.txt | tail -50
