#!/bin/bash
# round-4 GPU call J: why is the split-K fold 2.5x slower at B = 1 than in round 1?  PMC passes over both packages; HI arms at bench level
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for P in r1 head; do
  PK=""; [ $P = r1 ] && PK="--pkg $R/tools/_build_r1"
  i=0
  for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_${P}/p$i -o p -- python $R/tools/graph_latency.py $PK --batches 1 > $O/pmc_${P}_p$i.log 2>&1
  done
  find $O/pmc_$P -name "*agent*" -delete
  python - $O/pmc_$P <<'PY'
import csv, glob, sys, collections, re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        k=re.sub(r"\(.*","",r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ',''))
        if 'splitk' in k or 'gemm_g2' in k:
            agg[(k, r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X',''))][r['Counter_Name']].append(float(r['Counter_Value']))
print('==', sys.argv[1])
for k,cs in sorted(agg.items()):
    print(' ', k, {n: round(sum(v)/len(v),1) for n,v in sorted(cs.items())})
PY
done
find $O -name "*counter_collection.csv" -size +30M -delete; find $O -name "*kernel_trace.csv" -delete
cd $R
P=tools/_build_prod
KEEP=$O REPS=3 bash tools/ab_bench.sh cres0=${P}_cres0/libmetaenc.so hi12cres0=/tmp/cur.so 2>&1 | tee $O/ab_bench.txt
