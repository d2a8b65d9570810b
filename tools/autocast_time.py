"""Forward / train-step time of the Base encoder the way the reference's autocast recipes drive it (Video/engine_for_finetuning.py:92-99):
fp32 tokens and fp32 master weights under torch.autocast(bfloat16) -- bf16 MFMA compute, fp32 residual stream -- against the bench's
bf16-token form.   python tools/autocast_time.py [B N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import metatransformer_amd as M
from metatransformer_amd import ops
B, N = [int(v) for v in sys.argv[1:3]] if len(sys.argv) >= 3 else (256, 197)
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = M.build_encoder(12, 768, 12).to(dev)
for p in enc.parameters():
    if p.dim() == 2:
        torch.nn.init.normal_(p, std=0.02)
x32 = torch.randn(B, N, 768, device=dev)
gy = torch.randn(B, N, 768, device=dev) / (B * N)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def fwd_autocast():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        enc(x32)
xb = x32.bfloat16()
def fwd_bf16_tokens():
    for b in enc: b.compute_dtype = torch.bfloat16
    with torch.no_grad():
        enc(xb)
    for b in enc: b.compute_dtype = None
print(f"forward, autocast (fp32 tokens / residual stream): {timeit(fwd_autocast):7.3f} ms")
print(f"forward, bf16 tokens (bench form):                 {timeit(fwd_bf16_tokens):7.3f} ms")
xg = x32.clone().requires_grad_(True)
def train_autocast():
    for p in enc.parameters(): p.grad = None
    xg.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = enc(xg)
    y.backward(gy)
enc.train()
print(f"fwd + bwd, autocast:                               {timeit(train_autocast, 5):7.3f} ms")
ops.gemm_profile(True); train_autocast(); torch.cuda.synchronize()
recs = ops.gemm_profile_read(with_plan=True); ops.gemm_profile(False)
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for op, dt, m, n, k, ms, plan in recs:
    if op in (0, 1):
        key = ("NT" if op == 0 else "TN", n, k, plan & 15)
        agg[key][0] += 1; agg[key][1] += ms
for k, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(k, c, f"{1e3 * ms / c:8.1f} us avg")
