#!/bin/bash
# fold gpurun_out/prof_$ROUND (tools/prof_round.sh) into profiles/$ROUND_* (ROUND defaults to r06)
cd ${GRAFT_REPO_ROOT:-/root/repo}
RND=${ROUND:-r06}; export ROUND=$RND
P=gpurun_out/prof_$RND; T=$(find $P/train -name "*kernel_trace.csv" | head -1); F=$(find $P/fwd -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py $T 28 > profiles/${RND}_train_summary.txt; TS=$(find $P/train_serial -name "*kernel_trace.csv" | head -1); [ -n "$TS" ] && python tools/prof_summary.py $TS 28 > profiles/${RND}_train_serial_summary.txt; [ -n "$TS" ] && tail -1 $P/train_serial.json > profiles/${RND}_bench_train_serial_under_rocprof.json; python tools/prof_summary.py $F 16 > profiles/${RND}_fwd_summary.txt
cp $(find $P/train -name "*kernel_stats.csv" | head -1) profiles/${RND}_train_kernel_stats.csv; cp $(find $P/fwd -name "*kernel_stats.csv" | head -1) profiles/${RND}_fwd_kernel_stats.csv
python tools/pmc_summary.py > /dev/null 2>&1
for f in base large512_fwd large512_train large1568_fwd large1568_fwd_fp8attn large1568_train mixed fp32 fp32_3xbf16; do tail -1 $P/bench_$f.json > profiles/${RND}_bench_$f.json; done
tail -1 $P/train.json > profiles/${RND}_bench_train_under_rocprof.json; tail -1 $P/fwd.json > profiles/${RND}_bench_fwd_under_rocprof.json
cp $P/refshapes.json profiles/${RND}_refshapes.json; cp $P/refshapes.txt profiles/${RND}_refshapes.txt
for wl in large1568 large512; do TL=$(find $P/${wl}_train -name "*kernel_trace.csv" 2>/dev/null | head -1); [ -n "$TL" ] && python tools/prof_summary.py $TL 16 > profiles/${RND}_${wl}_train_summary.txt; done
