#!/bin/bash
# rocprofv3 PMC passes over a short bench.py run (separate passes, --kernel-trace only, per MI355X_MICROARCH.md):
# HBM traffic (FETCH_SIZE / WRITE_SIZE), MFMA busy, wave-cycle split.  Output: gpurun_out/pmc_bench/<pass>/
R=${GRAFT_REPO_ROOT:-/root/repo}
MODE=${MODE:-train}
OUT=$R/gpurun_out/pmc_bench_$MODE
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for P in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o p -- python $R/bench.py --steps 1 --warmup 1 --mode $MODE --no-cpu-baseline --no-fwd-leg > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
