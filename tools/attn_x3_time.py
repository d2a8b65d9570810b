"""me_attention_fwd_x3 against the exact-fp32 and the bf16 forward kernels on one shape:  python tools/attn_x3_time.py [B N H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metatransformer_amd import ops
B, N, H = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else (256, 197, 12)
hd = 64
dev = torch.device("cuda:0")
qkv = [torch.randn(B * N, 3 * H * hd, device=dev) for _ in range(3)]      # rotating inputs (Infinity Cache)
def timeit(fn, n=20):
    for i in range(3): fn(qkv[i % 3])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(qkv[i % 3])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
fl = 4.0 * B * H * N * N * hd
for name, fn in (("x3 (out + planes)", lambda t: ops.attention_fwd_x3(t, B, N, H, hd, 0.125, planes=True)),
                 ("x3 (planes only)", lambda t: ops.attention_fwd_x3(t, B, N, H, hd, 0.125, planes=True, want_out=False)),
                 ("x3 (out only)", lambda t: ops.attention_fwd_x3(t, B, N, H, hd, 0.125)),
                 ("exact fp32", lambda t: ops.attention_fwd(t, B, N, H, hd, 0.125, False))):
    us = timeit(fn)
    print(f"{name:20s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s (algorithmic)", flush=True)
qb = [t.bfloat16() for t in qkv]
us = timeit(lambda t: None) if False else None
for i in range(3): ops.attention_fwd(qb[i], B, N, H, hd, 0.125, False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20): ops.attention_fwd(qb[i % 3], B, N, H, hd, 0.125, False)
torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 20 * 1e6
print(f"{'bf16':20s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
