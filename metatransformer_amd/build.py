"""Build libmetaenc.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m metatransformer_amd.build [--force] [--dev]

Sources: metatransformer_amd/csrc/*.hip  ->  metatransformer_amd/libmetaenc.so
(the .so is git-ignored but travels to the GPU box with the snapshot).

--dev builds a second library, tools/_build/libmetaenc_dev.so, from the same sources with -DME_DEV: it additionally
exports me_dev_set() (kernel-family / debug switches for the A/B runs of tools/gemm_dev) and tools/_build/gemm_dev,
the torch-free bench driver.  The shipped library has no such switches.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmetaenc.so")
OBJ_DIR = os.path.join(HERE, "csrc", "_obj")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-value", "-ffp-contract=fast"]
# per-source extra flags.  attention_x3.hip: its MFMA accumulators are read and rescaled by VALU code every key chunk (online softmax);
# left to itself hipcc parks them in AGPRs and moves all 96 of them out and back per chunk (v_accvgpr_read / _write: a third of the
# loop's VALU instructions) -- the VGPR form of the MFMA keeps them where the VALU reads them (gfx950's register file is unified).
FILE_FLAGS = {"attention_x3.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.isfile(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _deps() -> list:
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))
                  + [os.path.join(os.path.dirname(HERE), "include", "metaenc.h")])


def needs_build() -> bool:
    if not os.path.isfile(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(f) > t for f in _deps())


DEV_DIR = os.path.join(os.path.dirname(HERE), "tools", "_build")
DEV_OUT = os.path.join(DEV_DIR, "libmetaenc_dev.so")


def build(force: bool = False, verbose: bool = True, dev: bool = False, variant: str = "", defines=()) -> str:
    """variant / defines (dev builds only): a further copy of the dev library, tools/_build_<variant>/, compiled with the extra
    -D switches (compile-time A/B arms such as the resident GEMM's cache policies: tools/r4_policy_builds.sh)"""
    dev_dir = DEV_DIR + ("_" + variant if variant else "")
    out = os.path.join(dev_dir, "libmetaenc_dev.so") if dev else OUT
    obj_dir = os.path.join(dev_dir, "_obj") if dev else OBJ_DIR
    flags = FLAGS + (["-DME_DEV"] + ["-D" + d for d in defines] if dev else [])
    if not dev and variant:
        # a PRODUCT-flavoured A/B arm (no -DME_DEV): tools/_build_prod_<variant>/libmetaenc.so, swapped in for the in-tree library by
        # the same-box A/B scripts (tools/ab_bench.sh); never loaded by the package itself
        dev_dir = os.path.join(os.path.dirname(HERE), "tools", "_build_prod_" + variant)
        out, obj_dir, flags = os.path.join(dev_dir, "libmetaenc.so"), os.path.join(dev_dir, "_obj"), FLAGS + ["-D" + d for d in defines]
    if not dev and not variant and not force and not needs_build():
        return OUT
    hipcc = _hipcc()
    os.makedirs(obj_dir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    headers = [f for f in _deps() if f.endswith(".h")]
    hdr_t = max(os.path.getmtime(h) for h in headers)

    def compile_one(src: str) -> str:
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        if (not force and os.path.isfile(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and os.path.getmtime(obj) > hdr_t):
            return obj
        cmd = [hipcc] + flags + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print("[metaenc build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out + ".tmp"] + objs
    if verbose:
        print("[metaenc build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    if dev:
        exe = os.path.join(dev_dir, "gemm_dev")
        cmd = [hipcc, "-O2", "-std=c++17", f"--offload-arch={ARCH}", os.path.join(os.path.dirname(HERE), "tools", "gemm_dev.hip"),
               "-o", exe, "-L" + dev_dir, "-lmetaenc_dev", "-Wl,-rpath,$ORIGIN", "-lpthread"]
        if verbose:
            print("[metaenc build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    # python -m metatransformer_amd.build [--force] [--dev [--variant NAME -DX=1 -DY=2 ...]]
    _variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    _defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    print(build(force="--force" in sys.argv, dev="--dev" in sys.argv, variant=_variant, defines=_defs))
