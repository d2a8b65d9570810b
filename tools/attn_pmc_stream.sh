#!/bin/bash
# PMC passes over tools/attn_time.py at a long sequence (default 32 1568 16 64): LDS activity / conflicts and MFMA / VALU busy of the
# streaming attention kernels (forward: attn_fwd_stream16, backward: attn_bwd_dq_stream16 + attn_bwd_dkdv_stream16).  Separate passes.
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE=${SHAPE:-32 1568 16 64}
OUT=$R/gpurun_out/attn_pmc_stream
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for P in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES" \
         "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o p -- python $R/tools/attn_time.py $SHAPE > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<'PY' > $OUT/summary.txt
import csv, glob, collections, os, re
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/attn_pmc_stream"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"attn_\w+", r["Kernel_Name"])
        if not m: continue
        agg[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("   %-32s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
    a = {c: sum(v) / len(v) for c, v in agg[k].items()}
    if "SQ_LDS_IDX_ACTIVE" in a and "SQ_BUSY_CU_CYCLES" in a:
        print("   -> LDS array active %.0f %% of busy CU cycles, bank conflicts %.0f %% of LDS cycles" % (100 * a["SQ_LDS_IDX_ACTIVE"] / a["SQ_BUSY_CU_CYCLES"], 100 * a.get("SQ_LDS_BANK_CONFLICT", 0) / a["SQ_LDS_IDX_ACTIVE"]))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "SQ_BUSY_CYCLES" in a:
        print("   -> MFMA busy %.0f %% (SIMD-cycles / 4 over SQ busy cycles)" % (100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / 4 / a["SQ_BUSY_CYCLES"]))
PY
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
