#!/bin/bash
# round profile: rocprofv3 --kernel-trace --stats over the bench command (train leg, and --mode fwd), the PMC passes, and
# the bench lines of the other BASELINE configurations.  Everything lands under gpurun_out/; tools/prof_summary.py and
# tools/pmc_summary.py fold it into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
RND=${ROUND:-r06}
O=$R/gpurun_out/prof_$RND
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fwd-leg > $O/train.json 2> $O/train.err
echo "train rc=$?"
ME_WGRAD_OVERLAP=0 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_serial -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fwd-leg > $O/train_serial.json 2> $O/train_serial.err
echo "train (serial order: ME_WGRAD_OVERLAP=0) rc=$?"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd -o t -- python $R/bench.py --steps 5 --warmup 2 --mode fwd --no-cpu-baseline > $O/fwd.json 2> $O/fwd.err
echo "fwd rc=$?"
# configs 5 and 3 (Large, 1568 / 512 tokens): kernel traces of the train step
for wl in large1568 large512; do
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${wl}_train -o t -- python $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-fwd-leg > $O/${wl}_train_under_rocprof.json 2> $O/${wl}_train.err
  echo "$wl train rc=$?"
done
find $O -name "*agent*" -delete
ME_WGRAD_OVERLAP=0 MODE=train bash $R/tools/pmc_bench.sh
MODE=fwd bash $R/tools/pmc_bench.sh
cd $R
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_base.json 2> $O/bench_base.err; echo "base rc=$?"
timeout 300 python bench.py --workload large512 --mode fwd --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_large512_fwd.json 2>> $O/bench_other.err; echo rc=$?
timeout 300 python bench.py --workload large512 --steps 6 --warmup 2 --no-cpu-baseline --no-fwd-leg > $O/bench_large512_train.json 2>> $O/bench_other.err; echo rc=$?
timeout 300 python bench.py --workload large1568 --mode fwd --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_large1568_fwd.json 2>> $O/bench_other.err; echo rc=$?
timeout 300 python bench.py --workload large1568 --mode fwd --attn-dtype fp8 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_large1568_fwd_fp8attn.json 2>> $O/bench_other.err; echo rc=$?
timeout 300 python bench.py --workload large1568 --steps 6 --warmup 2 --no-cpu-baseline --no-fwd-leg > $O/bench_large1568_train.json 2>> $O/bench_other.err; echo rc=$?
timeout 300 python bench.py --workload mixed --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_mixed.json 2>> $O/bench_other.err; echo rc=$?
timeout 600 python bench.py --dtype fp32 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_fp32.json 2>> $O/bench_other.err; echo rc=$?
timeout 600 python bench.py --dtype fp32 --fp32-mode 3xbf16 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_fp32_3xbf16.json 2>> $O/bench_other.err; echo rc=$?
timeout 900 python tools/refshapes.py --out $O/refshapes.json > $O/refshapes.txt 2>> $O/bench_other.err; echo rc=$?
tail -3 $O/bench_other.err
du -sh $R/gpurun_out/*
