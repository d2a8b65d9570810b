"""Small-batch forward latency (Base, N = 197, bf16), eager and hipGraph replay:  python tools/graph_latency.py [nofold] [--pkg DIR]
--pkg DIR: import metatransformer_amd from DIR instead of this repository (same-box A/B against an older snapshot: tools/_build_r1/
holds the round-1 package + library, built from `git archive 26b68fd`)."""
import os, sys, time
import torch
_pkg = sys.argv[sys.argv.index('--pkg') + 1] if '--pkg' in sys.argv else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _pkg)
import metatransformer_amd as M
print("package:", os.path.dirname(M.__file__), flush=True)
dev = torch.device('cuda:0')
torch.manual_seed(0)
enc = M.build_encoder(12, 768, 12).to(dev).eval()
for b in enc: b.compute_dtype = torch.bfloat16
if 'nofold' in sys.argv:
    for b in enc: b.fold_norm = False      # LayerNorm as its own kernel instead of folded into qkv / fc1
_bl = [int(v) for v in sys.argv[sys.argv.index('--batches') + 1].split(',')] if '--batches' in sys.argv else [1, 8, 32]
for B in _bl:
    x = torch.randn(B, 197, 768, device=dev).bfloat16()
    with torch.no_grad():
        for _ in range(3): y = enc(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): y = enc(x)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 50 * 1e3
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): yg = enc(x)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            yg = enc(x)
        g.replay(); torch.cuda.synchronize()
        ok = torch.equal(yg, y)
        t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 50 * 1e3
    print(f"B={B}: eager {eager:.3f} ms, hipGraph replay {graph:.3f} ms, identical={ok}", flush=True)
