"""Time the residual GEMMs of an fp32 token stream (bf16 operands, bias + fp32 residual -> fp32): python tools/gemm_one_f32res.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metatransformer_amd import ops
dev = torch.device("cuda:0")
M = 50432
for N, K in ((768, 768), (768, 3072)):
    a = [torch.randn(M, K, device=dev).bfloat16() for _ in range(3)]
    w = (0.05 * torch.randn(N, K, device=dev)).bfloat16()
    bias = torch.randn(N, device=dev)
    res32 = [torch.randn(M, N, device=dev) for _ in range(3)]
    resb = [r.bfloat16() for r in res32]
    def t(fn, n=20):
        for i in range(3): fn(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n): fn(i % 3)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    ops.gemm_profile(True)
    us32 = t(lambda i: ops.gemm(a[i], w, bias=bias, residual=res32[i], out_dtype=torch.float32))
    recs = ops.gemm_profile_read(with_plan=True); ops.gemm_profile(False)
    usb = t(lambda i: ops.gemm(a[i], w, bias=bias, residual=resb[i]))
    us0 = t(lambda i: ops.gemm(a[i], w, bias=bias, out_dtype=torch.float32))
    fl = 2.0 * M * N * K
    print(f"N={N} K={K}: fp32 residual+out {us32:7.1f} us ({fl/us32/1e6:6.0f} TF, plan {recs[-1][-1]:#x}) | bf16 residual+out {usb:7.1f} us | bias only fp32 out {us0:7.1f} us", flush=True)
