#!/bin/bash
# round 5, final tree: rocprofv3 --kernel-trace --stats of the bench command (train, train in serial order, forward) -> profiles/r05_{train,train_serial,fwd}_summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r05
rm -rf $O/train $O/train_serial $O/fwd; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fwd-leg > $O/train.json 2> $O/train.err; echo "train rc=$?"
ME_WGRAD_OVERLAP=0 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_serial -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fwd-leg > $O/train_serial.json 2> $O/train_serial.err; echo "serial rc=$?"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd -o t -- python $R/bench.py --steps 5 --warmup 2 --mode fwd --no-cpu-baseline > $O/fwd.json 2> $O/fwd.err; echo "fwd rc=$?"
find $O -name "*agent*" -delete
cd $R
mkdir -p gpurun_out/r5fp
T=$(find $O/train -name "*kernel_trace.csv" | head -1); TS=$(find $O/train_serial -name "*kernel_trace.csv" | head -1); F=$(find $O/fwd -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py $T 28 > gpurun_out/r5fp/r05_train_summary.txt
python tools/prof_summary.py $TS 28 > gpurun_out/r5fp/r05_train_serial_summary.txt
python tools/prof_summary.py $F 16 > gpurun_out/r5fp/r05_fwd_summary.txt
cp $(find $O/train -name "*kernel_stats.csv" | head -1) gpurun_out/r5fp/r05_train_kernel_stats.csv
cp $(find $O/fwd -name "*kernel_stats.csv" | head -1) gpurun_out/r5fp/r05_fwd_kernel_stats.csv
tail -1 $O/train.json > gpurun_out/r5fp/r05_bench_train_under_rocprof.json; tail -1 $O/train_serial.json > gpurun_out/r5fp/r05_bench_train_serial_under_rocprof.json; tail -1 $O/fwd.json > gpurun_out/r5fp/r05_bench_fwd_under_rocprof.json
rm -rf $O/train $O/train_serial $O/fwd
head -12 gpurun_out/r5fp/r05_fwd_summary.txt | cut -c1-130
head -8 gpurun_out/r5fp/r05_train_summary.txt | cut -c1-130
