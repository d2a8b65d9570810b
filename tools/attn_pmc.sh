#!/bin/bash
# PMC passes over tools/attn_trace (N = 197 attention forward / backward): LDS conflicts and activity, VALU / MFMA busy.
# Separate passes, --kernel-trace only.  Output: gpurun_out/attn_pmc/summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/attn_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for P in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
         "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o p -- $R/tools/attn_trace > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<'PY' > $OUT/summary.txt
import csv, glob, collections, os, re
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/attn_pmc"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"attn_\w+", r["Kernel_Name"]); k = m.group(0) if m else r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("   %-32s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
