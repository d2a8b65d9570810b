"""Static checks on the compiled resident NT GEMM kernels (gemm_g3r_kernel<EPI, PRE>), from the gfx950 assembly hipcc emits
for metatransformer_amd/csrc/gemm3.hip -- no GPU needed:

  1. no scratch access and no SGPR-spill lane traffic (v_readlane / v_writelane) inside an inner K-loop (a spill there costs
     every K-tile; around the loops it costs once per item);
  2. the register that receives a work ticket (g3r_draw: an asynchronous returning atomic whose result the kernel retires with a
     hand-counted s_waitcnt) is neither overwritten nor spilled between the atomic and the v_readfirstlane that consumes it --
     the compiler cannot see that the value is still in flight.

The persistent attention backward (attn_bwd_ring16_kernel, attention.hip) draws its items the same way -- issued at the head of
phase A, consumed at its end, ~300-600 instructions apart in a 1024-thread kernel at ~128 VGPRs -- and gets check 2 as well
(ADVICE r3): between the atomic and the v_readfirstlane the ticket register must not be written, spilled, or COPIED (a v_mov /
v_accvgpr_write of it would read the register before the atomic has returned).

    python tools/check_g3r_isa.py            # compiles to a temporary directory, prints one line per kernel, exit 1 on a finding
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "metatransformer_amd", "csrc", "gemm3.hip")
SRC_ATTN = os.path.join(ROOT, "metatransformer_amd", "csrc", "attention.hip")


def _regs(tok):
    out = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.add(int(a))
    return out


def compile_asm(workdir, src=SRC):
    from metatransformer_amd import build as me_build
    out = os.path.join(workdir, os.path.basename(src)[:-4] + ".s")
    cmd = [me_build._hipcc()] + me_build.FLAGS + ["-S", "--cuda-device-only", src, "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    return open(out).read()


def check(asm, kernel="gemm_g3r_kernel", check_loops=True):
    """kernel: substring of the mangled names to check; check_loops: also run check 1 (inner MFMA loops free of spills)"""
    findings, report = [], []
    names = [m.group(1) for m in re.finditer(r"^(_Z\w*" + kernel + r"\w*):", asm, re.M)]
    for nm in names:
        pos = asm.index("\n" + nm + ":")
        lines = asm[pos:asm.find(".end_amdhsa_kernel", pos)].splitlines()
        labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for i, l in enumerate(lines):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and labels.get(m.group(1), i) < i:
                loops.append((labels[m.group(1)], i))
        inner = [lp for lp in loops if lp[1] - lp[0] < 1500 and any("v_mfma" in x for x in lines[lp[0]:lp[1]])]
        spills = [i for i, l in enumerate(lines) if ("scratch_" in l or "v_readlane_b32" in l or "v_writelane_b32" in l)
                  and any(a <= i <= b for a, b in inner)]
        if spills and check_loops:
            findings.append(f"{nm}: {len(spills)} spill operations inside an inner K-loop (first at line {spills[0]})")
        draws = 0
        for i, l in enumerate(lines):
            m = re.search(r"global_atomic_add\s+v(\d+),", l)
            if not m:
                continue
            draws += 1
            reg, status = int(m.group(1)), "no consumer found"
            for j in range(i + 1, min(i + 4000, len(lines))):
                t = lines[j].strip()
                if not t or t.startswith((";", ".")):
                    continue
                ops = t.split(None, 1)
                if len(ops) < 2:
                    continue
                args = ops[1].split(",")
                if "v_readfirstlane_b32" in t and reg in _regs(",".join(args[1:])):
                    status = "ok"
                    break
                if ops[0].startswith("scratch_store") and reg in _regs(ops[1]):
                    status = f"spilled {j - i} instructions behind the atomic"
                    break
                if ops[0].startswith(("v_mov_b32", "v_accvgpr_write", "v_mov_b64")) and reg in _regs(",".join(args[1:])):
                    status = f"copied {j - i} instructions behind the atomic (reads the register before the result is back): {t[:60]}"
                    break
                is_store = ops[0].startswith(("buffer_store", "global_store", "ds_write", "s_"))
                if not is_store and "v_permlane" not in ops[0] and reg in _regs(args[0]):
                    status = f"overwritten {j - i} instructions behind the atomic: {t[:60]}"
                    break
            if status != "ok":
                findings.append(f"{nm}: ticket register v{reg}: {status}")
        scratch = sum("scratch_" in l for l in lines)
        if check_loops:
            report.append(f"{nm}: {len(inner)} inner K-loops clean, {draws} ticket draws intact, {scratch} scratch operations outside the loops")
        else:
            if draws == 0:
                findings.append(f"{nm}: no ticket draw found")
            report.append(f"{nm}: {draws} ticket draws intact, {scratch} scratch operations in the kernel")
    if not names:
        findings.append(f"no {kernel} instantiation found in the assembly")
    return report, findings


def main():
    sys.path.insert(0, ROOT)
    with tempfile.TemporaryDirectory() as d:
        asm = compile_asm(d)
        asm_attn = compile_asm(d, SRC_ATTN)
    report, findings = check(asm)
    r2, f2 = check(asm_attn, kernel="attn_bwd_ring16_kernel", check_loops=False)
    report, findings = report + r2, findings + f2
    print("\n".join(report))
    for f in findings:
        print("FINDING:", f)
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
