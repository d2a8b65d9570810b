"""Run a few launches of selected GEMM shapes (for rocprofv3 --pmc passes).  python tools/gemm_one.py qkv_fwd fc2_fwd --iters 3"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metatransformer_amd import ops, _capi
from gemm_bench import SHAPES

names = [a for a in sys.argv[1:] if not a.startswith("--")]
iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 3
dev = torch.device("cuda:0")
for name, op, m, n, k in SHAPES:
    if name not in names:
        continue
    g = torch.Generator().manual_seed(1)
    if op == "nt":
        a = torch.randn(m, k, generator=g).bfloat16().to(dev); b = (0.05 * torch.randn(n, k, generator=g)).bfloat16().to(dev)
        code, odt = _capi.ME_GEMM_NT, torch.bfloat16
    else:
        a = torch.randn(k, m, generator=g).bfloat16().to(dev); b = torch.randn(k, n, generator=g).bfloat16().to(dev)
        code, odt = _capi.ME_GEMM_TN, torch.float32
    for _ in range(iters):
        ops.gemm(a, b, op=code, out_dtype=odt)
    torch.cuda.synchronize()
print("done")
