"""CPU baseline leg of bench.py: the oracle restatement (oracle/block_oracle.py -- a "port", the reference's timm
dependency is not installable here) timed on the GPU box's host cores.  TEST/BENCH INFRASTRUCTURE ONLY."""
from __future__ import annotations

import os
import time

import torch

from . import block_oracle as bo


def time_encoder(depth=12, dim=768, heads=12, N=197, batch=8, backward=True, budget_s=15.0, seed=0):
    """Forward(+backward) of the restated encoder on `batch` synthetic samples, repeated until ~budget_s of CPU work
    (min 2 timed iterations after 1 warm-up).  Returns samples/s and what was run."""
    # cores actually usable by this process (cgroup / affinity aware; os.cpu_count() over-reports inside containers
    # and oversubscribing OpenMP threads stalls the run)
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    threads = max(1, min(threads, int(os.environ.get("METAENC_CPU_THREADS", "64"))))
    torch.set_num_threads(threads)
    sd = bo.make_encoder_state_dict(depth, dim, seed=seed)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, N, dim, generator=g)
    go = torch.randn(batch, N, dim, generator=g)

    def step():
        if backward:
            bo.encoder_forward_backward(x, sd, heads, go)
        else:
            with torch.no_grad():
                bo.encoder_forward(x, sd, heads)

    step()
    t0 = time.perf_counter()
    it = 0
    while True:
        step()
        it += 1
        el = time.perf_counter() - t0
        if it >= 2 and el >= budget_s:
            break
        if it >= 50:
            break
    return {
        "value": batch * it / el, "unit": "samples/s", "cores": threads, "kind": "port",
        "sample": f"oracle/block_oracle.py torch-CPU fp32 {'fwd+bwd' if backward else 'fwd'} of the {depth}L/{dim}d encoder on "
                  f"[{batch},{N},{dim}] tokens, {it} timed iterations ({el:.1f} s) after 1 warm-up",
    }


if __name__ == "__main__":
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--tokens", type=int, default=197)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--budget-s", type=float, default=12.0)
    a = ap.parse_args()
    print(json.dumps(time_encoder(a.depth, a.dim, a.heads, a.tokens, a.batch, not a.forward_only, a.budget_s)))
