"""Which weight-gradient launches pay under a CU hold, and how much?  Folds a rocprofv3 --kernel-trace csv of tools/contention.py (one R) into:
per wgrad kernel name, the launch durations split by "a cu_hog kernel was running during the launch" / "not".
    rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python tools/contention.py --cus 16 --reserve 16 --steps 4
    python tools/contention_trace.py OUT/.../t_kernel_trace.csv
"""
import csv
import sys
from collections import defaultdict

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
hogs = [(s, e) for s, e, k in rows if "cu_hog" in k]
print(f"{len(rows)} kernels, {len(hogs)} hog launches; hog durations (us): "
      + ", ".join(f"{(e - s) / 1e3:.0f}" for s, e in hogs[:12]) + (" ..." if len(hogs) > 12 else ""))
agg = defaultdict(lambda: [[], []])
for s, e, k in rows:
    if "g3tn" not in k and "g3r_kernel" not in k and "attn_bwd" not in k:
        continue
    name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    ov = 0
    for hs, he in hogs:
        lo, hi = max(s, hs), min(e, he)
        if hi > lo:
            ov += hi - lo
    agg[name][1 if ov > 0.2 * (e - s) else 0].append((e - s) / 1e3)
print(f"{'kernel':46s} {'n free':>7} {'mean us':>9} {'n held':>7} {'mean us':>9} {'max us':>8}")
for name, (free, held) in sorted(agg.items()):
    mf = sum(free) / len(free) if free else float('nan')
    mh = sum(held) / len(held) if held else float('nan')
    print(f"{name:46s} {len(free):7d} {mf:9.1f} {len(held):7d} {mh:9.1f} {max(held) if held else float('nan'):8.1f}")
