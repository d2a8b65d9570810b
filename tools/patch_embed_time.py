"""Patch embed: the gather inside the GEMM's operand stager (me_patch_embed / me_patch_embed_wgrad, csrc/patch_embed.hip) against the
two-pass route (me_patchify + me_gemm) it replaces, forward and weight gradient, same process, interleaved, inputs rotated through
more buffers than the 256 MiB Infinity Cache holds.

    python tools/patch_embed_time.py [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metatransformer_amd import _capi, ops  # noqa: E402

CASES = [
    ("image B=256 (config 2)", (256, 3, 224, 224), (1, 16, 16, 1, 16, 16)),
    ("image B=32", (32, 3, 224, 224), (1, 16, 16, 1, 16, 16)),
    ("video B=8 x 16 frames", (8, 3, 16, 224, 224), (2, 16, 16, 2, 16, 16)),
]


def timed(fn, iters):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    Cout = 768
    print(f"{'case':28s} {'pass':8s} {'two-pass us':>12s} {'(gather us)':>12s} {'fused us':>10s} {'x':>6s}  TF/s fused")
    for name, shape, geom in CASES:
        nbuf = max(2, int(600e6 // (2 * torch.Size(shape).numel())) + 1)
        xs = [torch.randn(shape, device=dev).to(dt) for _ in range(nbuf)]
        K = shape[1] * geom[0] * geom[1] * geom[2]
        w = (0.03 * torch.randn(Cout, K, device=dev)).to(dt)
        bias = torch.randn(Cout, device=dev)
        assert ops.patch_embed_fused(xs[0], geom, dt, Cout)
        cols, tps = ops.patchify(xs[0], *geom, dt)
        M = cols.shape[0]
        dys = [torch.randn(M, Cout, device=dev).to(dt) for _ in range(nbuf)]
        flops = 2.0 * M * K * Cout

        def two_fwd(i):
            c, _ = ops.patchify(xs[i % nbuf], *geom, dt)
            return ops.gemm(c, w, bias=bias)

        def gather(i):
            return ops.patchify(xs[i % nbuf], *geom, dt)

        def fused_fwd(i):
            return ops.patch_embed(xs[i % nbuf], w, bias, None, geom, 0, dt)

        def two_wgrad(i):
            c, _ = ops.patchify(xs[i % nbuf], *geom, dt)
            dw = ops.gemm(dys[i % nbuf], c, op=_capi.ME_GEMM_TN, out_dtype=torch.float32)
            return dw, ops.colsum(dys[i % nbuf])

        def fused_wgrad(i):
            return ops.patch_embed_wgrad(xs[i % nbuf], geom, dys[i % nbuf], torch.float32, True)

        tg = timed(gather, a.iters)
        for label, two, fus in (("forward", two_fwd, fused_fwd), ("wgrad", two_wgrad, fused_wgrad)):
            t2, tf = timed(two, a.iters), timed(fus, a.iters)
            t2b, tfb = timed(two, a.iters), timed(fus, a.iters)
            t2, tf = min(t2, t2b), min(tf, tfb)
            print(f"{name:28s} {label:8s} {t2:12.1f} {tg:12.1f} {tf:10.1f} {t2 / tf:6.2f}  {flops / tf * 1e-6:8.0f}", flush=True)
        del xs, dys, cols


if __name__ == "__main__":
    main()
