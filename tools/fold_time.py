"""Weight-gradient GEMMs (split-K wgrad kernel + fp32 slab fold) of the Base and Large encoders at the BASELINE token counts, timed per
kernel: run under rocprofv3 --kernel-trace (tools/runs/r5_run_fold.sh folds the trace).

    python tools/fold_time.py [--lib tools/_build_prod_X/libmetaenc.so]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metatransformer_amd import _capi, ops  # noqa: E402

if "--lib" in sys.argv:
    i = sys.argv.index("--lib")
    _capi.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]

# (tokens, out features, in features): dW[out, in] = dY[tokens, out]^T X[tokens, in]
SHAPES = [(65536, 1024, 4096), (65536, 4096, 1024), (65536, 3072, 1024), (65536, 1024, 1024),
          (50432, 768, 3072), (50432, 3072, 768), (50432, 2304, 768), (50432, 768, 768)]
dev = torch.device("cuda:0")
for T, M, N in SHAPES:
    dys = [torch.randn(T, M, device=dev).bfloat16() for _ in range(3)]
    xs = [torch.randn(T, N, device=dev).bfloat16() for _ in range(3)]
    dw = torch.zeros(M, N, device=dev)
    for i in range(12):
        ops.gemm(dys[i % 3], xs[i % 3], op=_capi.ME_GEMM_TN, out=dw, beta=1.0, want_colsum_a=False)
    torch.cuda.synchronize()
    del dys, xs
print("done")
