#!/bin/bash
# round 6, call T: weight gradients on the side stream vs serial order at the Large configurations (configs 3 and 5), same box, two repetitions
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6t
mkdir -p $O
cd $R
for rep in 1 2; do
  for wl in large1568 large512; do
    for arm in overlap serial; do
      FLAG=""; [ $arm = serial ] && FLAG="--no-wgrad-overlap"
      timeout 600 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-fwd-leg $FLAG > $O/${wl}_${arm}_$rep.json 2> $O/err.txt || { echo "$wl $arm failed"; tail -2 $O/err.txt; continue; }
      python - "$O/${wl}_${arm}_$rep.json" "$wl $arm" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ok=j.get("other_kernels") or {}
print(f"{sys.argv[2]:22s} {j['ms_per_step']:8.3f} ms  gemm {j['roofline']['avg_launch_us']:6.1f} us wgrad {j['roofline']['wgrad_kernel']['avg_launch_us']:6.1f}  attn f/b {ok['attention_fwd']['avg_launch_us']:6.1f}/{ok['attention_bwd']['avg_launch_us']:7.1f}  power {(j.get('power') or {}).get('mean_W')}")
PY
    done
  done
done
