"""Developer tool: race hunt for the persistent attention kernels -- many launches of the same problem (optionally next to a
second stream that keeps the memory system busy), every output compared bit for bit with the first launch's.
    python tools/attn_stress.py [iters]"""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from metatransformer_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
bad = 0
for (B, N, H, hd) in ((256, 197, 12, 64), (64, 100, 12, 48), (32, 1568, 16, 64), (64, 592, 12, 64), (200, 209, 3, 64), (128, 300, 12, 64)):
    C = H * hd
    qkv = torch.randn(B * N, 3 * C, device=dev).bfloat16()
    do = torch.randn(B * N, C, device=dev).bfloat16()
    scale = hd ** -0.5
    out0, lse0 = ops.attention_fwd(qkv, B, N, H, hd, scale, True)
    dq0 = ops.attention_bwd(qkv, out0, do, lse0, B, N, H, hd, scale)
    noise = torch.empty(64 << 20, device=dev)
    side = torch.cuda.Stream()
    n_f = n_b = 0
    for i in range(iters):
        if i % 3 == 0:
            with torch.cuda.stream(side):          # unrelated traffic on another stream
                noise.add_(1.0)
        out, lse = ops.attention_fwd(qkv, B, N, H, hd, scale, True)
        dq = ops.attention_bwd(qkv, out0, do, lse0, B, N, H, hd, scale)
        n_f += int(not (torch.equal(out, out0) and torch.equal(lse, lse0)))
        n_b += int(not torch.equal(dq, dq0))
    torch.cuda.synchronize()
    print(f"B={B} N={N} H={H} hd={hd}: {iters} launches, forward mismatches {n_f}, backward mismatches {n_b}", flush=True)
    bad += n_f + n_b
sys.exit(1 if bad else 0)
