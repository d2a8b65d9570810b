"""fold gpurun_out/r5tiny/attn_tiny_ab.txt (tools/runs/r5_run_tiny.sh) into one line per shape"""
import re
import sys
rows = [l.strip() for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r5tiny/attn_tiny_ab.txt")]
lib, d = None, {}
for l in rows:
    if l.startswith("=="):
        lib = "old" if "notiny" in l else "new"
        continue
    m = re.match(r"(fwd|bwd)\s+(B=\d+ N=\d+ H=\d+ hd=\d+(?: fp32)?):\s+([\d.]+) us", l)
    if m:
        d.setdefault(m.group(2), {})[(m.group(1), lib)] = float(m.group(3))
print("# shape                         tiled kernels -> attention_tiny.hip (one wave per 16 tokens), us per launch, same box")
for k, v in d.items():
    print("%-34s fwd %7.1f -> %7.1f   bwd %7.1f -> %7.1f" % (k, v[("fwd", "old")], v[("fwd", "new")], v[("bwd", "old")], v[("bwd", "new")]))
