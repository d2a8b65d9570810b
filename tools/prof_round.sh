#!/bin/bash
# round profile: rocprofv3 --kernel-trace --stats over the default bench command (train + fwd leg) and over --mode fwd,
# then the PMC passes.  Everything lands under gpurun_out/; tools/prof_collect.py folds it into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fwd-leg > $O/train.json 2> $O/train.err
echo "train rc=$?"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd -o t -- python $R/bench.py --steps 5 --warmup 2 --mode fwd --no-cpu-baseline > $O/fwd.json 2> $O/fwd.err
echo "fwd rc=$?"
find $O -name "*.csv" | head; ls -la $O/train $O/fwd
# the agent trace / big csvs are not needed back
find $O -name "*agent*" -delete
MODE=train bash $R/tools/pmc_bench.sh
MODE=fwd bash $R/tools/pmc_bench.sh
du -sh $R/gpurun_out/*
