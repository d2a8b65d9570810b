"""Probe (dev tool, round 6): what would hipGraph capture of the frozen-encoder step (forward + dL/dx) give at the reference's own batch sizes?
Captures `y = enc(x); y.backward(gy)` once with torch.cuda.graph on static tensors and replays it; prints eager vs replay ms per step and checks
that the replayed dL/dx equals the eager one bit for bit.    python tools/graph_train_probe.py [--dtype bf16|fp32]
VERDICT r5 item 7 asked for this capture; DESIGN section 7 item 5 predicted <= 1.5 % from the 98.6 % busy trace -- this measures it."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metatransformer_amd as M

SHAPES = [("timeseries_forecast", 32, 96), ("tabular", 32, 100), ("graph", 32, 160), ("xray", 16, 257), ("pointcloud_cls", 32, 257)]
dtype = torch.float32 if "--dtype" in sys.argv and sys.argv[sys.argv.index("--dtype") + 1] == "fp32" else torch.bfloat16
dev = torch.device("cuda:0")
C, H, L = 768, 12, 12
for name, B, N in SHAPES:
    torch.manual_seed(0)
    enc = M.build_encoder(L, C, H).to(dev)
    for p in enc.parameters():
        if p.dim() == 2:
            torch.nn.init.normal_(p, std=0.02)
        p.requires_grad_(False)
    for blk in enc:
        blk.compute_dtype = dtype
    enc.eval()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, N, C, generator=g).to(dev).to(dtype).requires_grad_(True)
    gy = (torch.randn(B, N, C, generator=g) / (B * N)).to(dev).to(dtype)

    def step():
        x.grad = None
        y = enc(x)
        y.backward(gy)

    def timeit(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    eager = timeit(step)
    ref = x.grad.clone()
    # capture on a side stream (torch's rule), after warm-up launches on it
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    try:
        graph = torch.cuda.CUDAGraph()
        x.grad = None
        with torch.cuda.graph(graph):
            y = enc(x)
            y.backward(gy)
        gx = x.grad
        replay = timeit(graph.replay)
        same = torch.equal(gx, ref)
        print(f"{name:20s} B={B} N={N} {str(dtype)[6:]}: eager {eager:.3f} ms  graph replay {replay:.3f} ms  ({100 * (replay / eager - 1):+.1f} %)  dL/dx identical: {same}", flush=True)
    except Exception as e:      # noqa: BLE001
        print(f"{name:20s}: capture failed: {type(e).__name__}: {str(e)[:200]}", flush=True)
