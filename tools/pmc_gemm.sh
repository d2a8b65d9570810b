#!/bin/bash
# rocprofv3 PMC passes over a few GEMM launches; results (csv) under gpurun_out/pmc/<pass>/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "(SQ|TCC|TCP|GRBM|TA)_[A-Z0-9_a-z]+" | sort -u > $OUT/counters.txt
SHAPES="${SHAPES:-qkv_fwd fc2_fwd qkv_wgrad}"
i=0
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_BF16" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
         "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o p -- python $R/tools/gemm_one.py $SHAPES --iters 2 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
ls -R $OUT | head -40
