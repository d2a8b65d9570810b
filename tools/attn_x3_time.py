"""me_attention_fwd_x3 / _bwd_x3 against the exact-fp32 and the bf16 kernels on one shape:  python tools/attn_x3_time.py [B N H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metatransformer_amd import _capi, ops
if "--lib" in sys.argv:          # an A/B arm of the library (python -m metatransformer_amd.build --variant NAME ...)
    i = sys.argv.index("--lib")
    _capi.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
B, N, H = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else (256, 197, 12)
hd = 64
dev = torch.device("cuda:0")
qkv = [torch.randn(B * N, 3 * H * hd, device=dev) for _ in range(3)]      # rotating inputs (Infinity Cache)
do = torch.randn(B * N, H * hd, device=dev)
def timeit(fn, n=20):
    for i in range(3): fn(i % 3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i % 3)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
fl = 4.0 * B * H * N * N * hd
for name, fn in (("fwd x3 (out + planes)", lambda i: ops.attention_fwd_x3(qkv[i], B, N, H, hd, 0.125, planes=True)),
                 ("fwd x3 (planes only)", lambda i: ops.attention_fwd_x3(qkv[i], B, N, H, hd, 0.125, planes=True, want_out=False)),
                 ("fwd x3 (out only)", lambda i: ops.attention_fwd_x3(qkv[i], B, N, H, hd, 0.125)),
                 ("fwd exact fp32", lambda i: ops.attention_fwd(qkv[i], B, N, H, hd, 0.125, False))):
    us = timeit(fn)
    print(f"{name:24s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s (algorithmic)", flush=True)
outs = [ops.attention_fwd_x3(t, B, N, H, hd, 0.125, need_lse=True) for t in qkv]
for name, fn in (("bwd x3", lambda i: ops.attention_bwd_x3(qkv[i], outs[i][0], do, outs[i][1], B, N, H, hd, 0.125)),
                 ("bwd exact fp32", lambda i: ops.attention_bwd(qkv[i], outs[i][0], do, outs[i][1], B, N, H, hd, 0.125))):
    us = timeit(fn, 10)
    print(f"{name:24s} {us:8.1f} us  {2.5 * fl / us / 1e6:7.1f} TF/s (algorithmic)", flush=True)
qb = [t.bfloat16() for t in qkv]
us = timeit(lambda i: ops.attention_fwd(qb[i], B, N, H, hd, 0.125, False))
print(f"{'fwd bf16':24s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
