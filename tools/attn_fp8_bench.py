"""fp8 vs bf16 attention forward on BASELINE config 5's shape (dev tool; run on the GPU box):
    python tools/attn_fp8_bench.py  ->  one JSON line with both timings, TFLOP/s against the 5 PF (fp8) / 2.5 PF (bf16) peaks"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metatransformer_amd import ops

B, N, H, hd = 32, 1568, 16, 64
dev = torch.device("cuda:0")
qkvs = [torch.randn(B * N, 3 * H * hd, device=dev).bfloat16() for _ in range(3)]
out = {}
for name, fp8 in (("bf16", False), ("fp8", True)):
    for _ in range(3):
        ops.attention_fwd(qkvs[0], B, N, H, hd, 0.125, need_lse=False, fp8=fp8)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        ops.attention_fwd(qkvs[i % 3], B, N, H, hd, 0.125, need_lse=False, fp8=fp8)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    fl = 4.0 * B * H * N * N * hd
    peak = 5000.0 if fp8 else 2500.0
    out[name] = {"us": round(us, 1), "TFLOPs": round(fl / us / 1e6, 1), "peak_TFLOPs": peak, "frac": round(fl / us / 1e6 / peak, 4)}
out["shape"] = {"B": B, "N": N, "H": H, "head_dim": hd, "note": "fp8 time includes the absmax + quantise/re-layout pre-pass"}
print(json.dumps(out))
