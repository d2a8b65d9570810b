#!/bin/bash
# round 6, final tree: the round profile (PMC on the final kernel sources) + the whole GPU suite + smoke()
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/runs/r6_run_prof.sh > gpurun_out/final_prof.log 2>&1; tail -8 gpurun_out/final_prof.log
mkdir -p gpurun_out/r6final
timeout 3000 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/r6final/gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/r6final/gpu_suite.txt
timeout 600 python __graft_entry__.py smoke > gpurun_out/r6final/smoke.txt 2>&1; echo "smoke rc=$?"; grep smoke gpurun_out/r6final/smoke.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r6final/bench_driver_command.json 2> gpurun_out/r6final/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r6final/bench_driver_command.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value","ms_per_step","mfma_frac_end_to_end")}, j.get("power"), j["roofline"]["frac"], j["roofline"]["traffic"], (j.get("fwd") or {}).get("ms_per_step"), (j.get("fwd") or {}).get("mfma_frac"), (j.get("fwd") or {}).get("power"))
PY
