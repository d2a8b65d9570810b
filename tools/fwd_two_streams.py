"""Experiment: the forward of the Base encoder at [256,197,768] bf16 as ONE chain on one stream against TWO independent half-batch
chains on two streams (the second chain's kernels fill the tails and launch gaps of the first's).  python tools/fwd_two_streams.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import metatransformer_amd as M

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, N, C, H, L = 256, 197, 768, 12, 12
enc = M.build_encoder(L, C, H).to(dev)
for p in enc.parameters():
    if p.dim() == 2:
        torch.nn.init.normal_(p, std=0.02)
enc.eval()
for blk in enc:
    blk.compute_dtype = torch.bfloat16
x = torch.randn(B, N, C, device=dev).bfloat16()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one():
    with torch.no_grad():
        return enc(x)


def split(parts):
    outs = []
    cur = torch.cuda.current_stream()
    streams = (s1, s2, torch.cuda.Stream(), torch.cuda.Stream())[:parts]
    for s in streams:
        s.wait_stream(cur)
    with torch.no_grad():
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                outs.append(enc(x[i * B // parts:(i + 1) * B // parts]))
    for s in streams:
        cur.wait_stream(s)
    return torch.cat(outs, 0)


ref = one()
for parts in (2,):
    got = split(parts)
    torch.cuda.synchronize()
    print(f"{parts} chains: identical to one chain: {torch.equal(ref, got)}  max diff {float((ref.float() - got.float()).abs().max()):.3e}")
for rep in range(3):
    for name, fn in (("one chain  [256]", one), ("two chains [128]x2", lambda: split(2)), ("four chains [64]x4", lambda: split(4))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        print(f"{name:20s} {1e3 * (time.perf_counter() - t0) / 10:7.3f} ms", flush=True)
