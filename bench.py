#!/usr/bin/env python
"""bench.py -- encoder samples/sec at B=256, N=197, C=768 (Base) on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode train|fwd] [--batch 256] [--no-cpu-baseline]
    (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

One "step" (mode=train, the default; BASELINE config 2 "Meta-Transformer-Base forward+backward ... [256,197,768], bf16"):
    forward of the 12-layer/768-d encoder on a per-GPU batch of synthetic bf16 tokens, backward (input + all weight
    gradients), one all-reduce per flat gradient bucket when N > 1, fused AdamW step on the fp32 master weights.
mode=fwd times the forward only (torch.no_grad), the configuration the north star's "40 % MFMA" target is quoted on.

Prints ONE JSON line (rank 0).  `value` = whole-job samples/s with the tokens resident in HBM.  The `roofline` object is
for the dominant kernel (the bf16 NT MFMA GEMM): algorithmic FLOPs of its launches / their summed duration, measured with
events on the launch stream inside the timed region.  `cpu_baseline` = the CPU oracle on the host cores (rank 0, N=1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md: ~2.5 PF dense)
PEAK_HBM_TBPS = 8.0             # HBM3E (same guide)
PEAK_HBM_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["train", "fwd"], default="train")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--tokens", type=int, default=197)
    ap.add_argument("--model", choices=["base", "large"], default="base")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("ME_BENCH_FORCE_DIST") == "1"     # (forced: exercises the RCCL path on 1 GPU)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    import metatransformer_amd as M
    from metatransformer_amd import ops, parallel, _capi

    L, C, H = (12, 768, 12) if args.model == "base" else (24, 1024, 16)
    B, N = args.batch, args.tokens
    torch.manual_seed(0)                         # identical weights on every rank
    enc = M.build_encoder(L, C, H).to(dev)
    for p in enc.parameters():                   # N(0, 0.02) weights in the checkpoint layout (no .pth available)
        if p.dim() == 2:
            torch.nn.init.normal_(p, std=0.02)
    for blk in enc:
        blk.compute_dtype = torch.bfloat16       # fp32 master weights, bf16 MFMA compute, bf16 token stream
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)        # per-rank data (Video/run_class_finetuning.py:417)
    x = torch.randn(B, N, C, generator=g).to(dev).bfloat16()
    gy = (torch.randn(B, N, C, generator=g) / (B * N)).to(dev).bfloat16()

    train = args.mode == "train"
    if train:
        enc.train()
        flat = parallel.FlatParams(enc.parameters())
        opt = parallel.FusedAdamW(flat, lr=1e-4, weight_decay=0.05)
        reducer = parallel.OverlappedGradReducer(flat, force=use_dist) if use_dist else None
        x.requires_grad_(True)                   # the tokenizer in front of the encoder needs dL/dx

        def step():
            flat.zero_grad()
            x.grad = None
            y = enc(x)
            y.backward(gy)                       # bucket all-reduces launch from grad hooks while backward still runs
            if reducer is not None:
                reducer.finish()
            opt.step(grad_scale=1.0 / world)
    else:
        enc.eval()

        def step():
            with torch.no_grad():
                enc(x)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    ops.gemm_profile(True)       # HIP events around every me_gemm launch, recorded by the library on the launch stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    prof = ops.gemm_profile_read()
    ops.gemm_profile(False)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel: bf16 NT MFMA GEMM (forward + dgrad launches share one kernel)
    nt = [(M_, N_, K_, ms_) for (op, dt, M_, N_, K_, ms_) in prof if op == _capi.ME_GEMM_NT and dt == _capi.ME_BF16]
    tn = [(M_, N_, K_, ms_) for (op, dt, M_, N_, K_, ms_) in prof if op == _capi.ME_GEMM_TN and dt == _capi.ME_BF16]
    roof = None
    if nt:
        flops = sum(2.0 * m * n * k for m, n, k, _ in nt)
        ms = sum(t for *_, t in nt)
        ach = flops / (ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "gemm_g2_kernel<BM,256,NT,EPI> (all bf16 NT MFMA GEMM launches: forward + dgrad)", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": None,
                "launches_per_step": len(nt) // args.steps, "avg_launch_us": round(1e3 * ms / len(nt), 2),
                "avg_launch_gflop": round(flops / len(nt) / 1e9, 2),
                "share_of_step_time": round(ms * 1e-3 / elapsed, 4)}
        # HBM bytes per launch of the same kernels from the committed rocprofv3 PMC pass (tools/pmc_bench.sh ->
        # tools/pmc_summary.py: 2*FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md correction); counters cannot be read live
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", f"r01_pmc_{args.mode}.json")))
            pmc_steps = pmc.get("_meta", {}).get("bench_steps_in_pass", 2)
            # every kernel an NT me_gemm call launches: the main kernel, its tail-split part (EPI 5) and that part's fold
            n_tail = sum(v["launches"] for k, v in pmc.items() if k.startswith("gemm_g2_kernel") and ", false, 5>" in k)
            tot_mb = sum(v["launches"] * v["hbm_traffic_MB"] for k, v in pmc.items()
                         if k.startswith("gemm_g2_kernel") and ", false," in k and "hbm_traffic_MB" in v)
            if "splitk_reduce_kernel" in pmc and n_tail:
                tot_mb += n_tail * pmc["splitk_reduce_kernel"]["hbm_traffic_MB"]
            if tot_mb > 0:
                roof["traffic"] = round(1e6 * tot_mb / (pmc_steps * (len(nt) // args.steps)))
                roof["traffic_unit"] = "HBM bytes per me_gemm(NT) call (PMC pass in profiles/, not live)"
                roof["algorithmic_bytes_per_launch"] = round(sum(2.0 * (m * k + n * k + m * n) for m, n, k, _ in nt) / len(nt))
        except Exception:      # noqa: BLE001 -- profile file absent: traffic stays null
            pass
        if tn:
            f2 = sum(2.0 * m * n * k for m, n, k, _ in tn)
            ms2 = sum(t for *_, t in tn)
            roof["wgrad_kernel"] = {"kernel": "gemm_g2_kernel<128,256,TN,5> (wgrad, split-K)", "achieved": round(f2 / (ms2 * 1e-3) / 1e12, 2),
                                    "unit": "TFLOP/s", "avg_launch_us": round(1e3 * ms2 / len(tn), 2),
                                    "share_of_step_time": round(ms2 * 1e-3 / elapsed, 4)}

    # ---- the other kernel classes of the step, from the same event records: HBM-bound LayerNorm against the 8 TB/s
    # peak (algorithmic bytes: SURVEY 8d), attention against both peaks
    def _class(code, work_fn, bytes_fn):
        recs = [(M_, N_, K_, ms_) for (op, dt, M_, N_, K_, ms_) in prof if op == code]
        if not recs:
            return None
        ms = sum(t for *_, t in recs)
        by = sum(bytes_fn(m, n, k) for m, n, k, _ in recs)
        out_ = {"launches_per_step": len(recs) // args.steps, "avg_launch_us": round(1e3 * ms / len(recs), 2),
                "hbm_TBps": round(by / (ms * 1e-3) / 1e12, 3), "hbm_frac": round(by / (ms * 1e-3) / 1e12 / PEAK_HBM_TBPS, 4),
                "share_of_step_time": round(ms * 1e-3 / elapsed, 4)}
        if work_fn is not None:
            fl = sum(work_fn(m, n, k) for m, n, k, _ in recs)
            out_["TFLOPs"] = round(fl / (ms * 1e-3) / 1e12, 1)
        return out_
    es = 2      # bf16 token stream
    other = {
        "layernorm_fwd": _class(_capi.ME_PROF_LN_FWD, None, lambda m, n, k: 2.0 * m * n * es),
        "layernorm_bwd": _class(_capi.ME_PROF_LN_BWD, None, lambda m, n, k: 4.0 * m * n * es),
        "attention_fwd": _class(_capi.ME_PROF_ATTN_FWD, lambda m, n, k: 4.0 * m * n * n * k, lambda m, n, k: 4.0 * m * n * k * es),
        "attention_bwd": _class(_capi.ME_PROF_ATTN_BWD, lambda m, n, k: 10.0 * m * n * n * k, lambda m, n, k: 8.0 * m * n * k * es),
    }
    other = {k: v for k, v in other.items() if v is not None}

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed
    fwd_flops = M.encoder_flops_per_sample(N, C, L)
    model_flops = (3.0 if train else 1.0) * fwd_flops
    out = {
        "metric": "encoder samples/sec at B=256,N=197,C=768 (Base)" if (args.model == "base" and B == 256 and N == 197)
                  else f"encoder samples/sec at B={B},N={N},C={C}",
        "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": (("BASELINE config 2: " if (args.model == "base" and B == 256 and N == 197) else "")
                                + f"Meta-Transformer-{args.model.capitalize()} "
                                + ("forward+backward+AdamW" if train else "encoder forward (no_grad)"))
                               + f", tokens [{B},{N},{C}] bf16 per GPU, "
                               f"{L}L/{C}d/{H}h, fp32 master weights, random init N(0,0.02)",
                   "mode": args.mode, "per_gpu_batch": B, "global_batch": B * world, "tokens": N,
                   "parallelism": f"dp{world}" if world > 1 else "single",
                   "grad_allreduce": "RCCL all-reduce(sum) per 64 MiB flat fp32 bucket, launched from grad hooks (overlaps backward), 1/world folded into AdamW" if world > 1 else None},
        "model_tflops_per_s": round(value * model_flops / 1e12, 2),
        "mfma_frac_end_to_end": round(value * model_flops / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
        "roofline": roof,
        "other_kernels": other,
    }
    if not args.no_cpu_baseline and world == 1:
        # the CPU oracle runs in its own process (fresh OpenMP pool, hard timeout) so it can never stall the bench
        import subprocess
        cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--depth", str(L), "--dim", str(C), "--heads", str(H),
               "--tokens", str(N), "--batch", "8", "--budget-s", str(args.cpu_budget_s)] + ([] if train else ["--forward-only"])
        try:
            r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=args.cpu_budget_s * 6 + 120)
            cb = json.loads(r.stdout.strip().splitlines()[-1])
            cb["value"] = round(cb["value"], 3)
            out["cpu_baseline"] = cb
        except Exception as e:          # noqa: BLE001 -- a baseline failure must not lose the GPU measurement
            out["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": None, "kind": "port",
                                   "sample": f"failed: {type(e).__name__}: {e}"[:300]}
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
