"""hipGraph capture of the encoder forward at a batch the resident GEMM takes (dev check; run on the GPU box):
warm up on a side stream, capture on torch's capture stream, replay, compare with eager."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import metatransformer_amd as M

dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = M.build_encoder(4, 768, 12).to(dev).eval()
for b in enc:
    b.compute_dtype = torch.bfloat16
x = torch.randn(128, 197, 768, device=dev).bfloat16()
with torch.no_grad():
    y = enc(x)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            enc(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        yg = enc(x)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("graph == eager:", torch.equal(yg, y), " second eager == first:", torch.equal(enc(x), y))
