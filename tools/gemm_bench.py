"""GEMM kernel-family A/B on the encoder's shapes (dev tool; run on the GPU box).
    python tools/gemm_bench.py [--quick]
Sets ME_GEMM_KERNEL per subprocess-free loop by re-importing is not possible (env is read once), so each family runs
in its own process: the parent spawns `python tools/gemm_bench.py --family X`."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

M = 256 * 197
SHAPES = [  # (name, op, M, N, K)   names ending in _gelu / _res / _aux use that fused epilogue
    ("fc1_fwd_gelu", "nt", M, 3072, 768), ("fc2_fwd_res", "nt", M, 768, 3072), ("proj_fwd_res", "nt", M, 768, 768),
    ("fc2_dgrad_aux", "nt", M, 3072, 768),
    ("qkv_fwd", "nt", M, 2304, 768), ("proj_fwd", "nt", M, 768, 768), ("fc1_fwd", "nt", M, 3072, 768),
    ("fc2_fwd", "nt", M, 768, 3072), ("fc1_dgrad", "nt", M, 768, 3072), ("fc2_dgrad", "nt", M, 3072, 768),
    ("qkv_dgrad", "nt", M, 768, 2304),
    ("qkv_wgrad", "tn", 2304, 768, M), ("proj_wgrad", "tn", 768, 768, M), ("fc1_wgrad", "tn", 3072, 768, M),
    ("fc2_wgrad", "tn", 768, 3072, M),
]


def run_family(family, iters):
    import torch
    from metatransformer_amd import ops, _capi
    dev = torch.device("cuda:0")
    res = {}
    NSET = 3                      # rotate operand / output buffers so the 256 MiB Infinity Cache cannot hold them (as in-model)
    for name, op, m, n, k in SHAPES:
        g = torch.Generator().manual_seed(1)
        if op == "nt":
            a0 = torch.randn(m, k, generator=g).bfloat16().to(dev)
            b = (0.05 * torch.randn(n, k, generator=g)).bfloat16().to(dev)
            As = [a0] + [a0.clone() for _ in range(NSET - 1)]
            Bs = [b] * NSET
            code = _capi.ME_GEMM_NT
            ref = lambda: a0[:512].float() @ b.float().t()
            sl = lambda c: c[:512]
        else:
            a0 = torch.randn(k, m, generator=g).bfloat16().to(dev)
            b0 = torch.randn(k, n, generator=g).bfloat16().to(dev)
            As = [a0] + [a0.clone() for _ in range(NSET - 1)]
            Bs = [b0] + [b0.clone() for _ in range(NSET - 1)]
            code = _capi.ME_GEMM_TN
            ref = lambda: a0.float().t() @ b0.float()
            sl = lambda c: c
        bias = torch.randn(n, device=dev)
        out_dtype = torch.bfloat16 if op == "nt" else torch.float32
        outs = [torch.empty((m, n), dtype=out_dtype, device=dev) for _ in range(NSET)]
        kw = dict(op=code, bias=bias if op == "nt" else None)
        extra = None
        if name.endswith("_gelu"):
            kw["act"] = _capi.ME_ACT_GELU
        elif name.endswith("_res"):
            extra = torch.randn(m, n, device=dev).bfloat16(); kw["residual"] = extra
        elif name.endswith("_aux"):
            extra = torch.randn(m, n, device=dev).bfloat16(); kw["aux"] = extra; kw["bias"] = None
        c = ops.gemm(As[0], Bs[0], out=outs[0], **kw)
        r = ref() + (bias if (op == "nt" and kw.get("bias") is not None) else 0)
        if name.endswith("_gelu"):
            r = torch.nn.functional.gelu(r)
        elif name.endswith("_res"):
            r = r + extra[:512].float()
        elif name.endswith("_aux"):
            xa = extra[:512].float()
            r = r * (0.5 * (1 + torch.erf(xa / 2 ** 0.5)) + xa * torch.exp(-0.5 * xa * xa) / (2 * 3.141592653589793) ** 0.5)
        err = float((sl(c).float() - r).abs().max() / r.abs().max())
        for i in range(3):
            ops.gemm(As[i % NSET], Bs[i % NSET], out=outs[i % NSET], **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            ops.gemm(As[i % NSET], Bs[i % NSET], out=outs[i % NSET], **kw)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / iters
        res[name] = {"us": round(us, 1), "tflops": round(2.0 * m * n * k / us / 1e6, 1), "relerr": float(f"{err:.2e}")}
        del As, Bs, outs
    print(json.dumps({"family": family, "results": res}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--family", default=None)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--families", default="g128,g256_256,g256_128,auto")
    a = ap.parse_args()
    if a.family is not None:
        run_family(a.family, a.iters)
        sys.exit(0)
    rows = []
    for fam in a.families.split(","):           # "family[:variant]"
        env = dict(os.environ)
        base, _, rest = fam.partition(":")
        var, _, dbg = rest.partition(":")
        if dbg:
            env["ME_G256_DEBUG"] = dbg
        if base != "auto":
            env["ME_GEMM_KERNEL"] = base
        else:
            env.pop("ME_GEMM_KERNEL", None)
        if var:
            env["ME_G256_VARIANT"] = var
        r = subprocess.run([sys.executable, __file__, "--family", fam, "--iters", str(a.iters)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(fam, "FAILED", r.stderr[-2000:])
            continue
        rows.append(json.loads(line[-1]))
    names = [s[0] for s in SHAPES]
    print(f"{'shape':12s} " + " ".join(f"{r['family']:>16s}" for r in rows))
    for n in names:
        print(f"{n:12s} " + " ".join(f"{r['results'][n]['tflops']:7.1f}TF {r['results'][n]['relerr']:.0e}" for r in rows))
    tot = [sum(r['results'][n]['us'] for n in names) for r in rows]
    print(f"{'sum us':12s} " + " ".join(f"{t:16.1f}" for t in tot))
