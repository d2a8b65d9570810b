"""CPU baseline leg of bench.py: the encoder on the GPU box's host cores.  TEST/BENCH INFRASTRUCTURE ONLY.

What is timed is the reference's own CPU formulation -- `Block.forward` as the stock fused ATen calls the reference makes
(F.layer_norm, F.linear, softmax, F.gelu: PointCloud/openpoints/models/layers/attention.py:26-38,55-58, mlp.py:30-35) --
a "port" (the reference's timm dependency is not installable here).  tests/test_oracle.py pins it to the explicit
restatement in block_oracle.py.  Thread count and batch are SWEPT (more threads is not faster on a 12-layer / 768-d
model at batch 8; larger batches feed more cores) and the best configuration is reported with its core count.
"""
from __future__ import annotations

import os
import time

import torch
import torch.nn.functional as F

from . import block_oracle as bo


def block_forward_fused(x, sd, i, heads, eps=1e-5):
    """One Block exactly as the reference evaluates it on CPU (fused ATen ops)."""
    B, N, C = x.shape
    hd = C // heads
    p = lambda k: sd[f"{i}.{k}"]          # noqa: E731
    h = F.layer_norm(x, (C,), p("norm1.weight"), p("norm1.bias"), eps)
    qkv = F.linear(h, p("attn.qkv.weight"), p("attn.qkv.bias")).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
    a = (attn @ v).transpose(1, 2).reshape(B, N, C)
    x = x + F.linear(a, p("attn.proj.weight"), p("attn.proj.bias"))
    h = F.layer_norm(x, (C,), p("norm2.weight"), p("norm2.bias"), eps)
    h = F.gelu(F.linear(h, p("mlp.fc1.weight"), p("mlp.fc1.bias")))
    return x + F.linear(h, p("mlp.fc2.weight"), p("mlp.fc2.bias"))


def encoder_forward_fused(x, sd, heads, eps=1e-5):
    depth = 1 + max(int(k.split(".")[0]) for k in sd)
    for i in range(depth):
        x = block_forward_fused(x, sd, i, heads, eps)
    return x


def _usable_cores() -> int:
    try:
        return len(os.sched_getaffinity(0))       # cgroup / affinity aware
    except AttributeError:
        return os.cpu_count() or 1


def time_encoder(depth=12, dim=768, heads=12, N=197, backward=True, budget_s=20.0, seed=0, replicas=False):
    """Sweep (threads, batch) and report the best samples/s with the configuration that gave it.

    Threads ascend through {8, 16, 32, 64, all usable}; a leg that is SLOWER than the best so far ends the ascent (throughput
    of this 12-layer / 768-d model is unimodal in the thread count: on a 256-core host the 64- and 256-thread legs used to burn
    most of the bench's wall time to report 4.9 and 0.1 samples/s).  Batches {8, 32, 64} are then timed at the best thread
    count and, for the larger batches, at the next count up (bigger GEMMs feed more cores), again stopping at the first leg that
    does not improve.  Every leg is one warm-up + timed iterations; the whole sweep stays inside `budget_s` (soft)."""
    cores = _usable_cores()
    cand = sorted({c for c in (8, 16, 32, 64, cores) if c <= cores} or {cores})
    sd = bo.make_encoder_state_dict(depth, dim, seed=seed)
    if backward:
        sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    g = torch.Generator().manual_seed(seed)
    data = {}

    def make(batch):
        if batch not in data:
            data[batch] = (torch.randn(batch, N, dim, generator=g), torch.randn(batch, N, dim, generator=g))
        return data[batch]

    def step(x, go):
        if backward:
            xr = x.clone().requires_grad_(True)
            y = encoder_forward_fused(xr, sd, heads)
            torch.autograd.grad(y, [xr] + list(sd.values()), go)
        else:
            with torch.no_grad():
                encoder_forward_fused(x, sd, heads)

    t_start = time.perf_counter()
    legs = {}                                  # (threads, batch) -> (samples/s, iterations, seconds)

    def leg(th, batch, min_iters=1, max_s=4.0):
        torch.set_num_threads(th)
        x, go = make(batch)
        step(x, go)                            # warm-up (thread pool, allocator)
        t0, it = time.perf_counter(), 0
        while True:
            step(x, go)
            it += 1
            el = time.perf_counter() - t0
            if it >= min_iters and (el >= max_s or it >= 20 or time.perf_counter() - t_start > sweep_s):
                break
        legs[(th, batch)] = (batch * it / el, it, el)
        return legs[(th, batch)][0]

    # with --replicas the single-process sweep gets 60 % of the budget and the replica leg the rest.  Measured on the 256-core host of
    # an MI355X box (round 4): 16 replicas x 16 threads = 7.8 samples/s in aggregate against 11.6 for ONE 16-thread process -- the
    # host is memory-bound long before its cores are used, and starting 16 interpreters costs 80 s -- so the replica leg is opt-in
    # and the one-process figure is the reported baseline.
    sweep_s = (0.6 if replicas else 1.0) * budget_s
    asc_s = max(1.5, 0.1 * budget_s)           # per leg of the ascent: the 8 / 16 / 32-thread legs all get their turn
    best_th, best = cand[0], leg(cand[0], 8, max_s=asc_s)
    for th in cand[1:]:
        if time.perf_counter() - t_start > 0.5 * sweep_s:
            break
        v = leg(th, 8, max_s=asc_s)
        if v <= best:
            break                               # slower than the best so far: stop ascending
        best_th, best = th, v
    up = [c for c in cand if c > best_th][:1]
    for batch in (32, 64):
        if time.perf_counter() - t_start > 0.85 * sweep_s:
            break
        improved = False
        for th in [best_th] + up:
            if time.perf_counter() - t_start > sweep_s:
                break
            v = leg(th, batch, max_s=max(2.0, 0.15 * budget_s))
            if v > best:
                best, improved = v, True
        if not improved:
            break
    (th_b, batch_b), (val, it, el) = max(legs.items(), key=lambda kv: kv[1][0])
    sweep = ", ".join(f"{t}x{b}: {v[0]:.1f}" for (t, b), v in sorted(legs.items()))
    # What the HOST can do: one process of this model stops scaling at 8 - 32 threads, so R = cores / threads independent replicas of
    # the best single-process configuration run side by side, each pinned to its own cores (one 16-thread process uses
    # 6 % of a 256-core box).  The aggregate is the reported value when it is the larger one.
    rep = None
    R = min(cores // th_b, 32)
    left = budget_s - (time.perf_counter() - t_start)
    if R >= 2 and replicas and left > 4.0:
        rep = _run_replicas(R, th_b, batch_b, depth, dim, heads, N, backward, min(6.0, max(3.0, 0.6 * left)))
    head = (f"reference Block formulation (fused ATen ops, oracle/cpu_baseline.py) torch-CPU fp32 {'fwd+bwd' if backward else 'fwd'} of the "
            f"{depth}L/{dim}d encoder; host has {cores} usable cores")
    if rep is not None and rep["value"] > val:
        return {"value": rep["value"], "unit": "samples/s", "cores": R * th_b, "kind": "port",
                "sample": (f"{head}; {R} independent replicas x {th_b} threads (pinned core sets) on [{batch_b},{N},{dim}] tokens each, "
                           f"{rep['iters']} iterations in {rep['seconds']:.1f} s, aggregate of the replicas' own rates; best single process "
                           f"{val:.1f} samples/s at {th_b} threads; sweep (threads x batch: samples/s): {sweep}; "
                           f"{time.perf_counter() - t_start:.0f} s in all")}
    return {
        "value": val, "unit": "samples/s", "cores": th_b, "kind": "port",
        "sample": (f"{head}; one process on [{batch_b},{N},{dim}] tokens, {it} timed iterations ({el:.1f} s) at {th_b} of {cores} cores; "
                   f"sweep (threads x batch: samples/s): {sweep}; ascent stops at the first slower leg"
                   + (f"; {R} replicas x {th_b} threads gave {rep['value']:.1f} samples/s in aggregate" if rep is not None else "")
                   + f"; {time.perf_counter() - t_start:.0f} s in all"),
    }


def _replica_worker(th, batch, depth, dim, heads, N, backward, secs, cpus):
    """one replica: pinned to `cpus`, `th` threads, steps for `secs` seconds after a warm-up; prints its own rate"""
    import json
    if cpus:
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            pass
    torch.set_num_threads(th)
    sd = bo.make_encoder_state_dict(depth, dim, seed=0)
    if backward:
        sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    g = torch.Generator().manual_seed(0)
    x, go = torch.randn(batch, N, dim, generator=g), torch.randn(batch, N, dim, generator=g)

    def step():
        if backward:
            xr = x.clone().requires_grad_(True)
            y = encoder_forward_fused(xr, sd, heads)
            torch.autograd.grad(y, [xr] + list(sd.values()), go)
        else:
            with torch.no_grad():
                encoder_forward_fused(x, sd, heads)
    step()
    t0, it = time.perf_counter(), 0
    while time.perf_counter() - t0 < secs:
        step()
        it += 1
    el = time.perf_counter() - t0
    print(json.dumps({"rate": batch * it / el, "iters": it, "seconds": el}), flush=True)


def _run_replicas(R, th, batch, depth, dim, heads, N, backward, secs):
    import json
    import subprocess
    import sys
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        avail = list(range(os.cpu_count() or 1))
    procs = []
    for r in range(R):
        cpus = avail[r * th:(r + 1) * th]
        cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--worker", str(th), str(batch), str(secs), ",".join(map(str, cpus)),
               "--depth", str(depth), "--dim", str(dim), "--heads", str(heads), "--tokens", str(N)] + ([] if backward else ["--forward-only"])
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                      cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    rate, iters, worst = 0.0, 0, 0.0
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=secs * 6 + 120)
            j = json.loads(out.strip().splitlines()[-1])
            rate += j["rate"]; iters += j["iters"]; worst = max(worst, j["seconds"])
        except Exception:       # noqa: BLE001 -- a replica that failed simply does not count
            pr.kill()
    return {"value": rate, "iters": iters, "seconds": worst} if rate > 0 else None


if __name__ == "__main__":
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--tokens", type=int, default=197)
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--budget-s", type=float, default=20.0)
    ap.add_argument("--batch", type=int, default=0, help="(ignored: the batch is swept)")
    ap.add_argument("--replicas", action="store_true", help="also time cores / threads pinned replicas of the best configuration")
    ap.add_argument("--worker", nargs=4, default=None, metavar=("THREADS", "BATCH", "SECONDS", "CPUS"), help="(internal: one replica)")
    a = ap.parse_args()
    if a.worker:
        th, batch, secs, cpus = int(a.worker[0]), int(a.worker[1]), float(a.worker[2]), [int(c) for c in a.worker[3].split(",") if c]
        _replica_worker(th, batch, a.depth, a.dim, a.heads, a.tokens, not a.forward_only, secs, cpus)
    else:
        print(json.dumps(time_encoder(a.depth, a.dim, a.heads, a.tokens, not a.forward_only, a.budget_s, replicas=a.replicas)))
