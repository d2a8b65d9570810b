#!/bin/bash
# round-5 GPU call I: PMC counters of the x3 attention forward kernel (own run, --kernel-trace + --pmc only)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O/p1 -o p -- python $R/tools/attn_x3_time.py > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/p2 -o p -- python $R/tools/attn_x3_time.py > $O/p2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for p in sorted(glob.glob('gpurun_out/r5i/p*/**/p_counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        if 'x3' in r['Kernel_Name'] or 'attn_fwd' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in agg.items():
        print(k, {n: round(sum(v) / len(v)) for n, v in cs.items()}, 'launches', max(len(v) for v in cs.values()))
PY
