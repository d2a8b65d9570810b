"""Summarise a rocprofv3 --kernel-trace csv: per (kernel, grid) count / avg / total / share.  usage: prof_summary.py trace.csv [top]"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
agg = collections.defaultdict(list)
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    agg[(name[:64], r['Grid_Size_X'], r['Grid_Size_Y'])].append(d)
tot = sum(sum(v) for v in agg.values())
print(f"total kernel time {tot/1e3:.2f} ms over {len(rows)} dispatches")
print(f"{'kernel':64s} {'grid':>14s} {'n':>5s} {'avg_us':>9s} {'total_ms':>9s} {'share':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{k[0]:64s} {k[1]+'x'+k[2]:>14s} {len(v):5d} {sum(v)/len(v):9.1f} {sum(v)/1e3:9.2f} {100*sum(v)/tot:5.1f}%")
