"""Fold the rocprofv3 --pmc passes of tools/pmc_bench.sh into one JSON per mode (per-kernel averages per launch).
HBM traffic = 2 * FETCH_SIZE + WRITE_SIZE  (KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced
streams -- MI355X_MICROARCH.md "HBM" -- hence the factor 2; WRITE_SIZE is uncalibrated and taken as reported)."""
import collections, csv, glob, hashlib, json, os, re, sys


def _head():
    """commit the profile was taken on: git where there is a checkout, else tools/_build/HEAD (written on the build box right before
    the gpurun call -- the GPU box gets a snapshot without .git)"""
    h = os.popen("git rev-parse --short HEAD 2>/dev/null").read().strip()
    if not h:
        try:
            h = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "HEAD")).read().strip()
        except OSError:
            h = ""
    return h or None

def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name)

ROUND = os.environ.get("ROUND", "r06")
# the sources that define the roofline kernel (gemm_g3r_kernel): bench.py prints `traffic: null` when the running tree's differ from
# the ones these counters were collected on (a committed constant silently going stale)
KERNEL_SRC = ["gemm3.hip", "gemm3_core.h", "gemm_common.h", "gemm.hip", "common.h"]


def kernel_src_sha16(root="."):
    h = hashlib.sha256()
    for f in KERNEL_SRC:
        with open(os.path.join(root, "metatransformer_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def main(mode):
    base = f"gpurun_out/pmc_bench_{mode}"
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for p in sorted(glob.glob(f"{base}/p*/p_counter_collection.csv")):
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for r in csv.DictReader(open(f"{base}/p1/p_kernel_trace.csv")):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for k, cs in agg.items():
        c = {n: sum(v) / len(v) for n, v in cs.items()}
        e = {"launches": len(dur.get(k, [])), "avg_us_profiled": round(sum(dur[k]) / max(1, len(dur[k])), 1)}
        if "FETCH_SIZE" in c:
            e["fetch_MB"] = round(2 * c["FETCH_SIZE"] / 1024, 1)
            e["write_MB"] = round(c.get("WRITE_SIZE", 0) / 1024, 1)
            e["hbm_traffic_MB"] = round(e["fetch_MB"] + e["write_MB"], 1)
        # (GRBM_GUI_ACTIVE over the launch's wall time is NOT used as a clock estimate here: the counter is summed over the eight XCDs
        # and over a different pass than the durations -- it gave 2.1 .. 2.7 "GHz".  The effective clock under a kernel is measured
        # inside it instead: s_memtime against s_memrealtime, tools/gemm_dev debug bit 8 -> profiles/r03_gemm_dev_clock.txt.)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            e["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 256 * 4), 4)
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            e["wave_cycles"] = {"wait_any": round(c["SQ_WAIT_ANY"] / wc, 3), "wait_inst": round(c["SQ_WAIT_INST_ANY"] / wc, 3),
                                "active": round(c["SQ_ACTIVE_INST_ANY"] / wc, 3)}
        if "TCC_HIT_sum" in c:
            e["l2_hit"] = round(c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3)
        out[k] = e
    out["_meta"] = {"bench_steps_in_pass": 2, "note": "tools/pmc_bench.sh runs bench.py --steps 1 --warmup 1: launch counts cover 2 steps",
                    "kernel_src_sha16": kernel_src_sha16(), "kernel_src": KERNEL_SRC,
                    "head": _head()}
    json.dump(out, open(f"profiles/{ROUND}_pmc_{mode}.json", "w"), indent=1, sort_keys=True)
    for k, e in sorted(((k, e) for k, e in out.items() if k != "_meta"), key=lambda kv: -kv[1]["launches"] * kv[1]["avg_us_profiled"])[:12]:
        print(f"{k[:60]:60s}", {x: e[x] for x in e if x != "wave_cycles"})

if __name__ == "__main__":
    for m in sys.argv[1:] or ["fwd", "train"]:
        print("==", m); main(m)
