"""The shapes the REFERENCE actually drives through the encoder, in the mode it drives them: frozen encoder,
forward + dL/dx only (a trainable tokenizer in front, `requires_grad=False` on every Block parameter), fp32 (the reference's
default arithmetic: README.md:113-150, PointCloud/.../metatransformer.py:144-169, Time-Series/models/MetaTransformer.py:80-88,
Hyper-spectrum/metatransformer.py:146-165) and bf16.  Per shape: ms per step, samples/s, model TFLOP/s against the dtype's MFMA
peak, and per kernel family the launches, mean microseconds, TFLOP/s or TB/s and -- for the GEMMs -- the plan the library chose
(me_gemm_profile_rec.plan).  VERDICT r3 item 4b.

    python tools/refshapes.py [--out profiles/r04_refshapes.json] [--quick] [--dtypes fp32,fp32x3,bf16]
(fp32x3 = fp32 tokens and weights with Block.fp32_mode = "3xbf16": fp32-accurate on the bf16 matrix pipe; its peak is 2 500 / 3 TF)
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import metatransformer_amd as M  # noqa: E402
from metatransformer_amd import _capi, ops  # noqa: E402

# name, per-GPU batch, tokens, C, heads, depth, source
SHAPES = [
    ("pointcloud_cls", 32, 257, 768, 12, 12, "PointCloud/cfgs/modelnet40ply2048: 1 + 1024 / 4 tokens, batch 32, fp32, frozen"),
    ("timeseries_forecast", 32, 96, 768, 12, 12, "Time-Series ETTh1: seq_len 96, batch 32, fp32, frozen"),
    ("hyperspectral", 64, 201, 768, 12, 12, "Hyper-spectrum Indian Pines: 200 bands + cls, batch 64, fp32, frozen"),
    ("graph_pcqm4m", 128, 50, 768, 32, 12, "Graph PCQM4Mv2: ~50 node + edge tokens, 32 heads (head_dim 24), batch 128, frozen"),
    ("pointcloud_s3dis", 8, 1501, 768, 12, 12, "PointCloud S3DIS segmentation: 1 + 24 000 / 16 tokens, batch 8, fp32, frozen"),
    ("xray_image", 32, 197, 768, 12, 12, "X-Ray / image classification: 196 patches + cls, batch 32, fp32, frozen"),
    ("tabular", 256, 16, 768, 12, 12, "Tabular Adult / Bank: ~14-20 column tokens, batch 256, fp32, frozen"),
    ("audio_sc", 32, 400, 768, 12, 12, "Audio Speech Commands: 20 x 20 tokens, 128 / 4 GPUs, fp16 autocast (bf16 here), frozen"),
]
PEAK = {torch.float32: 157.3, torch.bfloat16: 2500.0}
FAMILY = {0: "g128", 2: "g2b", 3: "g2w", 4: "g3"}


def run_shape(name, B, N, C, H, L, dtype, steps, warmup, fp32_mode="exact"):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    enc = M.build_encoder(L, C, H).to(dev)
    for p in enc.parameters():
        if p.dim() == 2:
            torch.nn.init.normal_(p, std=0.02)
        p.requires_grad_(False)                       # frozen encoder
    for blk in enc:
        blk.compute_dtype = dtype
        blk.fp32_mode = fp32_mode                     # "3xbf16": fp32-accurate arithmetic on the bf16 matrix pipe (Block.fp32_mode)
    enc.eval()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, N, C, generator=g).to(dev).to(dtype).requires_grad_(True)
    gy = (torch.randn(B, N, C, generator=g) / (B * N)).to(dev).to(dtype)

    def step():
        x.grad = None
        y = enc(x)
        y.backward(gy)                                # dL/dx through the frozen encoder: dgrad GEMMs, attention / LN backward, no wgrad

    def fwd():
        with torch.no_grad():
            enc(x)

    out = {}
    for label, fn, flop_mul in (("fwd_dx", step, 3.0 - 1.0), ("fwd", fwd, 1.0)):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if os.environ.get("REFSHAPES_STEP_TIMES") == "1":          # (diagnostic: every step synchronised and printed)
            ts = []
            for _ in range(steps):
                t1 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append(round(1e3 * (time.perf_counter() - t1), 3))
            print(f"    {name} {label} per-step ms: {ts}", flush=True)
        else:
            for _ in range(steps):
                fn()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
        ops.gemm_profile(True)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        recs = ops.gemm_profile_read(with_plan=True)
        ops.gemm_profile(False)
        # frozen: forward F + dgrad F (no wgrad) = 2 F for fwd + dx
        flops = M.encoder_flops_per_sample(N, C, L) * flop_mul * B
        fam = {}
        for op, dt, m, n, k, ms, plan in recs:
            if op in (_capi.ME_GEMM_NT, _capi.ME_GEMM_TN):
                key = f"gemm_{'nt' if op == _capi.ME_GEMM_NT else 'tn'}:{FAMILY.get(plan & 15, plan & 15)}" + (f"+splitk{plan >> 8}" if plan & 16 else "")
                e = fam.setdefault(key, {"launches": 0, "ms": 0.0, "flop": 0.0})
                e["flop"] += 2.0 * m * n * k
            else:
                key = {_capi.ME_PROF_LN_FWD: "layernorm_fwd", _capi.ME_PROF_LN_BWD: "layernorm_bwd", _capi.ME_PROF_ATTN_FWD: "attention_fwd",
                       _capi.ME_PROF_ATTN_BWD: "attention_bwd", _capi.ME_PROF_ROW_STATS: "row_stats"}.get(op, str(op))
                e = fam.setdefault(key, {"launches": 0, "ms": 0.0, "flop": 0.0})
                if op == _capi.ME_PROF_ATTN_FWD:
                    e["flop"] += 4.0 * m * n * n * k
                elif op == _capi.ME_PROF_ATTN_BWD:
                    e["flop"] += 10.0 * m * n * n * k
            e["launches"] += 1
            e["ms"] += ms
        fams = {k: {"launches_per_step": v["launches"] // 3, "avg_us": round(1e3 * v["ms"] / v["launches"], 1),
                    "TFLOPs": round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flop"] else None,
                    "share_of_kernel_time": round(v["ms"] / sum(q["ms"] for q in fam.values()), 3)} for k, v in sorted(fam.items())}
        out[label] = {"ms_per_step": round(1e3 * el, 3), "samples_per_s": round(B / el, 1),
                      "model_TFLOPs": round(flops / el / 1e12, 1),
                      "frac_of_mfma_peak": round(flops / el / 1e12 / (PEAK[torch.bfloat16] / 3 if fp32_mode == "3xbf16" else PEAK[dtype]), 4),
                      "kernel_ms_per_step": round(sum(v["ms"] for v in fam.values()) / 3, 3), "kernels": fams}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--dtypes", default="fp32,fp32x3,bf16")
    a = ap.parse_args()
    res = {"_meta": {"mode": "frozen encoder (requires_grad=False on every Block parameter): forward + dL/dx; `fwd` = forward alone under no_grad",
                     "peaks_TFLOPs": {"fp32": 157.3, "fp32x3": 833.3, "bf16": 2500.0}, "device": torch.cuda.get_device_name(0)}}
    for name, B, N, C, H, L, src in SHAPES:
        if a.only and a.only not in name:
            continue
        res[name] = {"source": src, "B": B, "N": N, "C": C, "heads": H, "depth": L}
        for dtype, dn in ((torch.float32, "fp32"), (torch.float32, "fp32x3"), (torch.bfloat16, "bf16")):
            if dn not in a.dtypes.split(","):
                continue
            steps = 5 if (dn == "fp32" or a.quick) else 20
            r = run_shape(name, B, N, C, H, L, dtype, steps, 2 if dn == "fp32" else 4, "3xbf16" if dn == "fp32x3" else "exact")
            res[name][dn] = r
            print(f"{name:20s} {dn}: fwd+dx {r['fwd_dx']['ms_per_step']:8.3f} ms ({r['fwd_dx']['frac_of_mfma_peak']:.3f} of peak)  fwd {r['fwd']['ms_per_step']:8.3f} ms "
                  f"({r['fwd']['frac_of_mfma_peak']:.3f})  gemm plans: {[k for k in r['fwd']['kernels'] if k.startswith('gemm')]}", flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
