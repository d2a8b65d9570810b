"""Measured ceiling only (never on the product path): torch.matmul -> hipBLASLt/rocBLAS on the encoder's GEMM shapes,
with rotating buffers like tools/gemm_bench.py."""
import torch
dev = torch.device("cuda:0")
M = 256 * 197
def t(f, n=12):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for i in range(n): f(i)
    e1.record(); torch.cuda.synchronize(); return 1e3 * e0.elapsed_time(e1) / n
for name, m, n, k in [("qkv_fwd", M, 2304, 768), ("proj_fwd", M, 768, 768), ("fc1_fwd", M, 3072, 768), ("fc2_fwd", M, 768, 3072)]:
    As = [torch.randn(m, k, device=dev).bfloat16() for _ in range(3)]
    W = (0.05 * torch.randn(n, k, device=dev)).bfloat16()
    b = torch.randn(n, device=dev).bfloat16()
    outs = [torch.empty(m, n, device=dev, dtype=torch.bfloat16) for _ in range(3)]
    us = t(lambda i=0: torch.addmm(b, As[i % 3], W.t(), out=outs[i % 3]))
    print(f"{name:10s} vendor addmm: {us:7.1f} us  {2.0*m*n*k/us/1e6:7.1f} TF")
name, T, m, n = "fc1_wgrad", M, 3072, 768
A = torch.randn(T, m, device=dev).bfloat16(); B = torch.randn(T, n, device=dev).bfloat16()
us = t(lambda i=0: torch.matmul(A.t(), B))
print(f"{name:10s} vendor A^T B : {us:7.1f} us  {2.0*T*m*n/us/1e6:7.1f} TF")
