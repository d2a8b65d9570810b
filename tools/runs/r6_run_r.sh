#!/bin/bash
# round 6, call R: the streaming backward with the 32-key dK / dV kernel -- race hunt, same-box A/B of the attention entry points against the
# 16-row arm, bench lines of configs 3 / 5 (Large, 512 and 1568 tokens)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6r
mkdir -p $O
cd $R
timeout 900 python tools/attn_stress.py 200 > $O/race.txt 2>&1; echo "race rc=$?"; cat $O/race.txt | tail -12
PREV=tools/_build_prod_rows16/libmetaenc.so bash tools/ab_attn_shapes.sh "32 1568 16 64" "128 592 12 64" "64 1000 12 64" "128 520 16 64" "16 3136 16 64" "64 1568 12 64" "64 577 12 64" > $O/ab.txt 2>&1
cat $O/ab.txt
for wl in large1568 large512; do
  timeout 600 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-fwd-leg > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?"
done
python - <<'PY'
import json
for wl in ("large1568","large512"):
    j=json.loads(open(f"gpurun_out/r6r/bench_{wl}.json").read().strip().splitlines()[-1])
    print(wl, j["ms_per_step"], {k:(v.get("avg_launch_us")) for k,v in (j.get("other_kernels") or {}).items()})
PY
