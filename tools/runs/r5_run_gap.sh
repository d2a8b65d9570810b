cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5gap; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python $GRAFT_REPO_ROOT/tools/refshapes.py --only timeseries --dtypes bf16 > $O/out.txt 2>&1
tail -1 $O/out.txt
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows=[(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last 20 'fwd' steps are no_grad; find the fwd_dx timed region: take a window in the middle of the trace with ln_bwd kernels
bw=[i for i,r in enumerate(rows) if "ln_bwd" in r[2]]
i0,i1=bw[len(bw)//4], bw[3*len(bw)//4]
seg=rows[i0:i1]
busy=sum(e-s for s,e,_ in seg); span=seg[-1][1]-seg[0][0]
gaps=[seg[i+1][0]-seg[i][1] for i in range(len(seg)-1)]
gaps.sort()
import collections, re
agg=collections.defaultdict(lambda:[0,0])
for st,en,k in seg:
    k=k.replace("(anonymous namespace)::","").replace("void ",""); k=re.sub(r"\(.*","",k)
    agg[k][0]+=1; agg[k][1]+=en-st
for k,(n,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:16]:
    print(f"  {k[:70]:70s} n={n:5d} avg {t/n/1e3:7.1f} us  {100*t/busy:5.1f} %")
print(f"fwd+dx window: {len(seg)} kernels, busy {busy/1e6:.3f} ms of {span/1e6:.3f} ms = {busy/span:.3f}; median gap {gaps[len(gaps)//2]/1e3:.1f} us, p90 {gaps[int(len(gaps)*0.9)]/1e3:.1f} us, mean {sum(gaps)/len(gaps)/1e3:.1f} us")
PY
rm -rf $O/t
