"""Developer tool: time the attention forward / backward entry points through the product library (no stamps) on rotating
buffers.    python tools/attn_time.py [--lib tools/_build_prod_X/libmetaenc.so] [B N H hd]"""
import os
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from metatransformer_amd import _capi, ops

if "--lib" in sys.argv:          # an A/B arm of the library (python -m metatransformer_amd.build --variant NAME -D...)
    i = sys.argv.index("--lib")
    _capi.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
DT = torch.bfloat16
if "--fp32" in sys.argv:
    sys.argv.remove("--fp32")
    DT = torch.float32
B, N, H, hd = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (256, 197, 12, 64)
dev = torch.device("cuda:0")
C = H * hd
NB = 6      # rotating buffers: B * N * 3C * 2 bytes = 232 MB each at the default shape (beyond the 256 MiB Infinity Cache together)
qkvs = [torch.randn(B * N, 3 * C, device=dev).to(DT) for _ in range(NB)]
douts = [torch.randn(B * N, C, device=dev).to(DT) for _ in range(NB)]
scale = hd ** -0.5
out, lse = ops.attention_fwd(qkvs[0], B, N, H, hd, scale, True)
ops.attention_bwd(qkvs[0], out, douts[0], lse, B, N, H, hd, scale)
torch.cuda.synchronize()
for name, fn in (("fwd", lambda i: ops.attention_fwd(qkvs[i % NB], B, N, H, hd, scale, True)),
                 ("fwd (no lse)", lambda i: ops.attention_fwd(qkvs[i % NB], B, N, H, hd, scale, False)),
                 ("bwd", lambda i: ops.attention_bwd(qkvs[i % NB], out, douts[i % NB], lse, B, N, H, hd, scale))):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 30
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / iters
    flops = 4.0 * B * H * N * N * hd * (1 if name.startswith("fwd") else 2.5)
    print(f"{name:14s} B={B} N={N} H={H} hd={hd}{' fp32' if DT == torch.float32 else ''}: {us:8.1f} us   {flops / us / 1e6:7.1f} TF/s")
