#!/usr/bin/env python
"""bench.py -- encoder samples/sec at B=256, N=197, C=768 (Base) on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode train|fwd] [--workload base|large512|large1568|mixed]
                    [--batch B] [--no-cpu-baseline]

--gpus N > 1 launches itself: when no torchrun environment is present the script re-executes under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, one rank per GPU; started
by torchrun directly (the driver's form) it reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.

One "step" (mode=train, the default; BASELINE config 2 "Meta-Transformer-Base forward+backward ... [256,197,768], bf16"):
    forward of the 12-layer/768-d encoder on a per-GPU batch of synthetic bf16 tokens, backward (input + all weight
    gradients), one RCCL all-reduce per flat gradient bucket when N > 1 (through the C ABI: me_allreduce_bucket on the
    communicator's own stream, launched from gradient hooks while backward still runs), fused AdamW on the fp32 masters.
The default run also times the encoder FORWARD alone (torch.no_grad) -- the configuration the north star's "40 % MFMA"
target is quoted on -- and reports it in the same JSON line under "fwd".

Prints ONE JSON line (rank 0).  `value` = whole-job samples/s with the tokens resident in HBM, timed with the library's
launch-timing hooks OFF.  `roofline` is for the dominant kernel (the bf16 NT MFMA GEMM): algorithmic FLOPs of its launches
/ their summed duration, from HIP events on the launch stream in a SEPARATE pass of the same steps right after the timed
region.  `cpu_baseline` = the reference's CPU formulation on the host cores (rank 0, N=1), threads and batch swept.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md: ~2.5 PF dense)
PEAK_F32_TFLOPS = 157.3        # f32-input MFMA (v_mfma_f32_32x32x2_f32): the f32 vector rate, 1/16 of bf16 (same guide)
PEAK_X3_TFLOPS = PEAK_BF16_TFLOPS / 3.0      # fp32-accurate mode: three bf16 MFMA products per fp32 product
PEAK_HBM_TBPS = 8.0            # HBM3E (same guide)

WORKLOADS = {       # name: (model, per-GPU batch, tokens)
    "base": ("base", 256, 197),          # BASELINE config 2 (the metric)
    "large512": ("large", 128, 512),     # config 3 token shape
    "large1568": ("large", 32, 1568),    # config 5 token shape
    "mixed": ("base", 64, 0),            # config 4: Image + Time-Series + Audio tokenizers, sequence-concat (README.md:122)
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["train", "fwd"], default="train")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="base")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (weak scaling); 0 = the workload's")
    ap.add_argument("--tokens", type=int, default=0)
    ap.add_argument("--model", choices=["base", "large"], default=None)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16",
                    help="fp32: the reference's default arithmetic (no autocast: README.md:113-150) -- fp32 tokens, exact-fp32 MFMA GEMMs "
                         "and attention; roofline against the 157 TF f32-input MFMA peak")
    ap.add_argument("--fp32-mode", choices=["exact", "3xbf16"], default="exact",
                    help="with --dtype fp32: 3xbf16 = fp32-accurate arithmetic on the bf16 matrix pipe (ME_BF16X3: three bf16 products per "
                         "Linear on hi / lo split operands, ~1e-5 relative; attention as three-product bf16 MFMA too -- me_attention_fwd_x3 / _bwd_x3 -- for head_dim 64, N > 64); roofline against 2500 / 3 TF")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fwd-leg", action="store_true")
    ap.add_argument("--no-wgrad-overlap", action="store_true",
                    help="train: keep the weight-gradient GEMMs in program order on the one stream (me_block_bwd_overlap(0))")
    ap.add_argument("--opt-prefetch", action="store_true",
                    help="A/B arm: rebuild the transposed weight copies on a side stream behind the optimizer step (under the next forward)")
    ap.add_argument("--opt-overlap", action="store_true",
                    help="A/B arm (single GPU): per-Block AdamW launches + gradient zero-fill + weight transposes on the optimizer's side stream "
                         "(FusedAdamW(overlap=True)) instead of one pass behind backward")
    ap.add_argument("--no-chain-stats", action="store_true",
                    help="A/B: folded inference reads every LayerNorm input once more (me_row_stats) instead of taking the statistics from the residual GEMMs' epilogues")
    ap.add_argument("--attn-dtype", choices=["bf16", "fp8"], default="bf16",
                    help="fp8: e4m3 attention forward on the block-scaled MFMA (config 5; head_dim 64)")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--grad-wire", choices=["fp32", "bf16"], default="fp32",
                    help="dtype of the gradient buckets on xGMI (bf16: rounded per bucket, summed by RCCL in bf16, restored to the "
                         "fp32 flat buffer before the optimizer; local accumulation stays fp32)")
    ap.add_argument("--mixed-per-rank", action="store_true",
                    help="workload mixed, second reading of BASELINE config 4: rank r tokenizes ONE modality (image / time series / "
                         "audio by r %% 3) instead of the README's sequence-concat of all three; encoder and gradient exchange shared")
    return ap.parse_args()


def self_launch(args) -> None:
    """--gpus N > 1 without a torchrun environment: start one rank per GPU and relay rank 0's JSON line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("ME_BENCH_FORCE_DIST") == "1"     # (forced: exercises the RCCL path on 1 GPU)

    import metatransformer_amd as M
    from metatransformer_amd import ops, parallel, _capi

    comm = None
    tgroup = None                # fallback only: a torch.distributed nccl (= RCCL) group
    comm_fallback = None
    if use_dist:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # host-side rendezvous only: carries the RCCL id to the ranks and the max-over-ranks of the timings
            dist.init_process_group("gloo")
            ok, why = 1, ""
            try:
                comm = parallel.Comm.from_torch_distributed()
            except Exception as e:       # noqa: BLE001 -- e.g. librccl not loadable outside torch's copy on this node
                ok, why = 0, f"{type(e).__name__}: {e}"
            flag = torch.tensor([ok])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # every rank takes the same path
            if int(flag.item()) == 0:
                if comm is not None:
                    comm.destroy()
                    comm = None
                tgroup = dist.new_group(backend="nccl")
                comm_fallback = f"torch.distributed nccl (RCCL) group -- the C-ABI communicator did not come up ({why or 'on another rank'})"
        else:
            comm = parallel.Comm(parallel.Comm.new_unique_id(), 0, 1)

    model, B0, N0 = WORKLOADS[args.workload]
    model = args.model or model
    L, C, H = (12, 768, 12) if model == "base" else (24, 1024, 16)
    B = args.batch or B0
    torch.manual_seed(0)                         # identical weights on every rank
    enc = M.build_encoder(L, C, H).to(dev)
    for p in enc.parameters():                   # N(0, 0.02) weights in the checkpoint layout (no .pth available)
        if p.dim() == 2:
            torch.nn.init.normal_(p, std=0.02)
    f32 = args.dtype == "fp32"
    x3 = f32 and args.fp32_mode == "3xbf16"
    tdt = torch.float32 if f32 else torch.bfloat16
    for blk in enc:
        blk.fp32_mode = "3xbf16" if x3 else "exact"
        blk.compute_dtype = tdt                  # fp32 master weights; bf16 MFMA compute on a bf16 token stream, or fp32 throughout
        blk.attn_fp8 = args.attn_dtype == "fp8"
        blk.chain_stats = not args.no_chain_stats
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)        # per-rank data (Video/run_class_finetuning.py:417)
    tok_note = None
    if args.workload == "mixed":
        # BASELINE config 4: three Data2Seq tokenizers feed one encoder; tokens are concatenated along the sequence
        # (the only multi-modal usage the reference shows: README.md:118-123)
        img = M.PatchEmbed(img_size=224, patch_size=16, in_c=3, embed_dim=C).to(dev)
        ts = M.DataEmbedding(c_in=7, d_model=C).to(dev).eval()
        aud = M.AcousticPatchEmbed(embed_dim=C).to(dev)
        with torch.no_grad():
            xi = img(torch.randn(B, 3, 224, 224, generator=g).to(dev))
            xt = ts(torch.randn(B, 96, 7, generator=g).to(dev))
            xa = aud(torch.randn(B, 1, 128, 256, generator=g).to(dev))
            if args.mixed_per_rank:
                # every rank feeds the shared encoder from a different tokenizer (sequence lengths differ per rank; nothing in
                # forward / backward depends on another rank, the gradient buckets have the same shape everywhere)
                x = (xi, xt, xa)[rank % 3].to(tdt).contiguous()
            else:
                x = torch.cat([xi, xt, xa], dim=1).to(tdt).contiguous()
        N = x.shape[1]
        tok_note = (f"Image 224/16 -> {xi.shape[1]} + Time-Series L96/c7 -> {xt.shape[1]} + Audio 128x256 k16/s10 -> {xa.shape[1]} tokens"
                    + ("; per-rank modality (rank r: modality r % 3), tokens = rank 0's" if args.mixed_per_rank else ""))
    else:
        N = args.tokens or N0
        x = torch.randn(B, N, C, generator=g).to(dev).to(tdt)
    gy = (torch.randn(B, N, C, generator=g) / (B * N)).to(dev).to(tdt)

    train = args.mode == "train"
    flat = opt = reducer = None
    if train:
        enc.train()
        # 1-D parameters and biases carry no weight decay, as in the reference recipes (Video/optim_factory.py:67-73)
        flat = parallel.FlatParams(enc.named_parameters(), no_decay=parallel.no_decay_rule)
        # (--opt-overlap: per-Block AdamW + zero-fill + weight transposes on the optimizer's side stream under backward / the next forward --
        #  measured SLOWER than the one-pass form, 30.4 vs 29.8 ms same box: profiles/r05_opt_overlap_ab.txt; the arm stays for A/B)
        opt = parallel.FusedAdamW(flat, lr=1e-4, weight_decay=0.05, overlap=args.opt_overlap and not use_dist, grad_scale=1.0 / world,
                                  prefetch_transposes=args.opt_prefetch)
        reducer = parallel.OverlappedGradReducer(flat, group=tgroup, comm=comm, force=use_dist,
                                                 wire_dtype=torch.bfloat16 if args.grad_wire == "bf16" else None) if use_dist else None
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

    def fwd_step():
        with torch.no_grad():
            enc(x)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    class _Power:
        """socket power over a timed region: hwmon power1_average, sampled every 10 ms from a host thread (rank 0; best effort: absent on
        boxes that do not expose it).  Why it is in the line: config 2 runs AT the socket's power cap, so joules per step is what sets
        the step time (profiles/r06_power.txt)."""

        def __init__(self):
            import glob
            self.files = (glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")
                          or glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")) if rank == 0 else []
            self.w, self.stop, self.th = [], False, None

        def _run(self):
            while not self.stop:
                best = 0.0
                for f in self.files:
                    try:
                        best = max(best, float(open(f).read().strip()) * 1e-6)
                    except (OSError, ValueError):
                        pass
                if best > 0:
                    self.w.append(best)
                time.sleep(0.01)

        def start(self):
            if self.files:
                import threading
                self.w, self.stop = [], False
                self.th = threading.Thread(target=self._run, daemon=True)
                self.th.start()

        def finish(self, seconds, steps):
            if self.th is None:
                return None
            self.stop = True
            self.th.join()
            self.th = None
            if len(self.w) < 3:
                return None
            mean = sum(self.w) / len(self.w)
            return {"mean_W": round(mean, 1), "max_W": round(max(self.w), 1), "samples": len(self.w),
                    "J_per_step": round(mean * seconds / steps, 2),
                    "source": "hwmon power1_average of the busiest visible socket, 10 ms samples over the timed steps (n_gpus = 1: this GPU)"}

    power_meter = _Power()
    power_info = {}

    def timed(fn, steps, warmup, power_key=None):
        for _ in range(warmup):
            fn()
        sync()
        if power_key:
            power_meter.start()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        el = time.perf_counter() - t0
        if power_key:
            power_info[power_key] = power_meter.finish(el, steps)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    def profiled(fn, steps):
        """the same steps again with the library's launch-timing hooks on (HIP events on the launch stream)"""
        ops.gemm_profile(True)
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        recs = ops.gemm_profile_read()
        ops.gemm_profile(False)
        return recs

    # The timed steps run the product's default schedule: me_block_bwd puts the weight-gradient GEMMs on its side stream, where they
    # fill the CUs the LayerNorm / attention backward kernels and the tails of the dY -> dX chain leave idle.  Kernels that share the
    # chip have no launch duration of their own, so the per-kernel records (roofline, other_kernels) come from a SERIAL pass of the
    # same step (me_block_bwd_overlap(0): every kernel alone on the chip, in program order) -- that pass is also timed, as
    # schedule.serial_ms_per_step.  `ME_WGRAD_OVERLAP=0 rocprofv3 ... bench.py` reproduces the serial durations, the plain command the
    # overlapped ones (profiles/r04_train_serial_summary.txt / r04_train_summary.txt).
    overlap = bool(train and not args.no_wgrad_overlap and os.environ.get("ME_WGRAD_OVERLAP", "1") != "0")
    ops.block_bwd_overlap(overlap)
    elapsed = timed(step, args.steps, args.warmup, power_key="step")
    psteps = max(1, min(args.steps, 5))
    serial_el = None
    if train:
        ops.block_bwd_overlap(False)
        if overlap:
            serial_el = timed(step, psteps, 1) / psteps
    prof = profiled(step, psteps)
    ops.block_bwd_overlap(overlap)
    fwd = None
    if train and not args.no_fwd_leg:
        enc.eval()
        fwd_el = timed(fwd_step, args.steps, min(args.warmup, 2), power_key="fwd")
        fwd_prof = profiled(fwd_step, psteps)
        enc.train()
        fwd = (fwd_el, fwd_prof)

    def pmc_traffic(mode):
        """HBM bytes per launch of the roofline kernel from the committed PMC passes of this same command (rocprofv3 --pmc
        cannot run inside the process: tools/pmc_bench.sh collects FETCH_SIZE / WRITE_SIZE in separate passes and
        tools/pmc_summary.py folds them, traffic = 2 * FETCH_SIZE + WRITE_SIZE).  None when the file is not there."""
        pm, src = None, None
        for rnd in ("r06", "r05", "r04", "r03"):            # the newest committed set
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"{rnd}_pmc_{mode}.json")
            try:
                with open(path) as f:
                    pm, src = json.load(f), f"profiles/{rnd}_pmc_{mode}.json"
                break
            except (OSError, ValueError):
                continue
        if pm is None:
            return None, None
        # stale? the file records the hash of the sources that define the roofline kernel at collection time
        meta = pm.get("_meta", {})
        try:
            import hashlib
            h = hashlib.sha256()
            for fn in meta.get("kernel_src", ["gemm3.hip", "gemm3_core.h", "gemm_common.h", "gemm.hip", "common.h"]):
                with open(os.path.join(ROOT, "metatransformer_amd", "csrc", fn), "rb") as fh:
                    h.update(fh.read())
            now = h.hexdigest()[:16]
        except OSError:
            now = None
        if meta.get("kernel_src_sha16") != now:
            return None, (f"{src} is STALE for this tree (collected on kernel sources {meta.get('kernel_src_sha16')}, head {meta.get('head')}; "
                          f"running {now}): re-run tools/pmc_bench.sh + tools/pmc_summary.py")
        rows = [(v["launches"], v["hbm_traffic_MB"]) for k, v in pm.items()
                if k.startswith("gemm_g3r_kernel") and isinstance(v, dict) and "hbm_traffic_MB" in v and v.get("launches")]
        if not rows:
            return None, None
        mib = sum(n * t for n, t in rows) / sum(n for n, _ in rows)        # (tools/pmc_summary.py reports MiB)
        return round(mib * 1048576), (f"{src}: rocprofv3 --pmc passes of this command (separate runs, 2*FETCH_SIZE + "
                                      f"WRITE_SIZE), launch-weighted mean over the gemm_g3r_kernel launches; not re-measured in this run")

    def aux_arrays(m, n, k):
        """how many extra [M, N] bf16 arrays the epilogue of this launch reads or writes (by the encoder's launch sequence)"""
        hid = 4 * C
        if n == hid:                       # fc1 (+ the saved gelu' in a training step) and the fc2 dgrad (* that saved factor)
            return 1.0 if train_now[0] else 0.0
        if n == C and k in (C, hid):       # forward proj / fc2: + residual.  A training step launches each of these two shapes
            return 0.5 if train_now[0] else 1.0      # twice, once as a forward (residual) and once as a dgrad (no row operand)
        return 0.0

    train_now = [False]                    # whether the record list being summarised comes from a training step

    def gemm_roofline(recs, wall_s, nsteps, with_wgrad):
        train_now[0] = bool(with_wgrad)
        cdt_code = _capi.ME_F32 if (f32 and not x3) else _capi.ME_BF16
        peak = PEAK_X3_TFLOPS if x3 else (PEAK_F32_TFLOPS if f32 else PEAK_BF16_TFLOPS)
        nt = [(M_, N_, K_, ms_) for (op, dt, M_, N_, K_, ms_) in recs if op == _capi.ME_GEMM_NT and dt == cdt_code]
        tn = [(M_, N_, K_, ms_) for (op, dt, M_, N_, K_, ms_) in recs if op == _capi.ME_GEMM_TN and dt == cdt_code]
        if not nt:
            return None
        if x3:      # a launch runs 3 K long rows (hi / lo planes): the ALGORITHMIC work is that of the fp32 Linear, 2 M N K
            nt = [(m, n, k // 3, t) for m, n, k, t in nt]
        flops = sum(2.0 * m * n * k for m, n, k, _ in nt)
        ms = sum(t for *_, t in nt)
        ach = flops / (ms * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic("train" if with_wgrad else "fwd") if args.workload == "base" and B == 256 and not f32 else (None, None)
        esz = 4.0 if f32 else 2.0
        roof = {"bound": "mfma",
                "kernel": ("gemm_g3_kernel<4> (bf16 NT MFMA GEMM over ME_BF16X3 three-plane operands, reduction 3 K, fp32 / three-plane output: every "
                           "forward + dgrad launch; flops counted as the fp32 Linear's 2 M N K, peak = 2500 / 3 TF)" if x3 else
                           "gemm_g128_kernel<float> (exact-fp32 NT GEMM on v_mfma_f32_32x32x2_f32, 128x128 tiles: every forward + dgrad launch)" if f32 else
                           "gemm_g3r_kernel<EPI> (bf16 NT MFMA GEMM, resident 256x256x64-tile workgroups: every forward + dgrad launch)"),
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "launches_per_step": len(nt) // nsteps, "avg_launch_us": round(1e3 * ms / len(nt), 2),
                "avg_launch_gflop": round(flops / len(nt) / 1e9, 2),
                # operands + output, 2 bytes each, plus the second [M, N] array the fused epilogues move: the residual read by
                # proj / fc2 (N = C), the saved gelu' written by fc1 in a training step and read by the fc2 dgrad (N or K = 4C)
                "algorithmic_bytes_per_launch": round(sum(esz * (m * k + n * k + m * n) + esz * m * n * aux_arrays(m, n, k) for m, n, k, _ in nt) / len(nt)),
                "share_of_step_time": round(ms * 1e-3 / nsteps / wall_s, 4)}
        if with_wgrad and tn:
            f2 = sum(2.0 * m * n * k for m, n, k, _ in tn)
            ms2 = sum(t for *_, t in tn)
            roof["wgrad_kernel"] = {"kernel": ("gemm_g3tn_x3_kernel + fold on ME_BF16X3 planes: ONE launch per weight gradient whose reduction runs over the three plane segments "
                                               "(hi,hi) / (lo,hi) / (hi,lo), bias gradient on it (csrc/gemm3_x3.hip); `achieved` counts the bf16 flops as launched (3x the fp32 Linear's)" if x3 else
                                               "gemm_g128_kernel<float, TN> (exact-fp32 wgrad)" if f32 else
                                               "gemm_g3tn_kernel + splitk_reduce_kernel (wgrad dW = dY^T X with the bias-gradient column sums fused; 256x256x64 tiles, split-K folded in fixed order)"),
                                    "achieved": round(f2 / (ms2 * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                                    "avg_launch_us": round(1e3 * ms2 / len(tn), 2),
                                    "share_of_step_time": round(ms2 * 1e-3 / nsteps / wall_s, 4)}
        return roof

    def other_kernels(recs, wall_s, nsteps):
        """the other kernel classes of the step, from the same event records: HBM-bound LayerNorm against the 8 TB/s
        peak (algorithmic bytes: SURVEY 8d), attention against both peaks"""
        def _class(code, work_fn, bytes_fn):
            rs = [(M_, N_, K_, ms_) for (op, dt, M_, N_, K_, ms_) in recs if op == code]
            if not rs:
                return None
            ms = sum(t for *_, t in rs)
            by = sum(bytes_fn(m, n, k) for m, n, k, _ in rs)
            o = {"launches_per_step": len(rs) // nsteps, "avg_launch_us": round(1e3 * ms / len(rs), 2),
                 "hbm_TBps": round(by / (ms * 1e-3) / 1e12, 3), "hbm_frac": round(by / (ms * 1e-3) / 1e12 / PEAK_HBM_TBPS, 4),
                 "share_of_step_time": round(ms * 1e-3 / nsteps / wall_s, 4)}
            if work_fn is not None:
                o["TFLOPs"] = round(sum(work_fn(m, n, k) for m, n, k, _ in rs) / (ms * 1e-3) / 1e12, 1)
            return o
        es = 4 if f32 else 2      # token stream
        d = {"row_stats": _class(_capi.ME_PROF_ROW_STATS, None, lambda m, n, k: 1.0 * m * n * es),
             "layernorm_fwd": _class(_capi.ME_PROF_LN_FWD, None, lambda m, n, k: 2.0 * m * n * es),
             "layernorm_bwd": _class(_capi.ME_PROF_LN_BWD, None, lambda m, n, k: 4.0 * m * n * es),
             "attention_fwd": _class(_capi.ME_PROF_ATTN_FWD, lambda m, n, k: 4.0 * m * n * n * k, lambda m, n, k: 4.0 * m * n * k * es),
             "attention_bwd": _class(_capi.ME_PROF_ATTN_BWD, lambda m, n, k: 10.0 * m * n * n * k, lambda m, n, k: 8.0 * m * n * k * es)}
        return {k: v for k, v in d.items() if v is not None}

    comm_info = comm.info() if comm is not None else None
    per_rank = None
    if world > 1:
        # per-rank averages of the two GEMM classes from the same event records: a communication kernel that holds CUs shows up
        # as slower NT / wgrad launches on that rank (the resident NT kernel claims its tiles, the wgrad kernel does not)
        def _avg(recs, code):
            v = [ms_ for (op, dt, M_, N_, K_, ms_) in recs if op == code and dt == _capi.ME_BF16]
            return round(1e3 * sum(v) / len(v), 2) if v else None
        mine = {"rank": rank, "tokens": int(N), "nt_avg_us": _avg(prof, _capi.ME_GEMM_NT), "wgrad_avg_us": _avg(prof, _capi.ME_GEMM_TN)}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    step_s = elapsed / args.steps
    value = world * B * args.steps / elapsed
    fwd_flops = M.encoder_flops_per_sample(N, C, L)
    model_flops = (3.0 if train else 1.0) * fwd_flops
    is_metric = args.workload == "base" and B == 256 and N == 197 and model == "base" and not f32
    what = "forward+backward+AdamW" if train else "encoder forward (no_grad)"
    out = {
        "metric": "encoder samples/sec at B=256,N=197,C=768 (Base)" if is_metric else f"encoder samples/sec at B={B},N={N},C={C}" + (" (fp32-accurate arithmetic, 3xbf16)" if x3 else " (fp32 arithmetic)" if f32 else ""),
        "value": round(value, 2), "unit": "samples/s", "n_gpus": comm_info["world"] if comm_info else world,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * step_s, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 as 3xbf16 (three bf16 MFMA products per Linear AND per attention product on hi / lo split operands, fp32 accumulate; softmax, LayerNorm, GELU, residual stream fp32)" if x3 else "f32" if f32 else ("bf16" if args.attn_dtype == "bf16" else "bf16 (attention forward in fp8 e4m3, fp32 softmax statistics)"),
        "data": "synthetic",
        "config": {"workload": ("BASELINE config 2: " if is_metric else ("BASELINE config 4 (sequence-concat): " if args.workload == "mixed" else ""))
                               + f"Meta-Transformer-{model.capitalize()} {what}, tokens [{B},{N},{C}] {'fp32' if f32 else 'bf16'} per GPU, "
                               f"{L}L/{C}d/{H}h, fp32 master weights, random init N(0,0.02)" + (f"; {tok_note}" if tok_note else ""),
                   "mode": args.mode, "per_gpu_batch": B, "global_batch": B * world, "tokens": N,
                   "parallelism": f"dp{world}" if world > 1 else "single",
                   "grad_allreduce": (f"me_allreduce_bucket (RCCL behind the C ABI, own stream, event hand-off): one all-reduce(sum) per "
                                      f"64 MiB flat fp32 bucket ({args.grad_wire} on the wire) launched from grad hooks, 1/world folded into AdamW; "
                                      f"{comm_info['buckets_reduced'] // max(1, args.steps + args.warmup + psteps)} buckets/step, "
                                      f"RCCL world {comm_info['world']}") if (comm_info and train) else
                                     (f"{comm_fallback}: one all_reduce(sum) per 64 MiB flat fp32 bucket from grad hooks" if (comm_fallback and train) else None)},
        "model_tflops_per_s": round(value * model_flops / 1e12, 2),
        "mfma_frac_end_to_end": round(value * model_flops / 1e12 / ((PEAK_X3_TFLOPS if x3 else PEAK_F32_TFLOPS if f32 else PEAK_BF16_TFLOPS) * world), 4),
        "per_rank_gemm": per_rank,
        "roofline": gemm_roofline(prof, serial_el or step_s, psteps, train),
        "other_kernels": other_kernels(prof, serial_el or step_s, psteps),
    }
    if power_info.get("step"):
        # (measured, not modelled: the step runs at the socket's power cap -- T = J / P)
        out["power"] = dict(power_info["step"], pJ_per_model_flop=round(power_info["step"]["J_per_step"] / (B * model_flops) * 1e12, 3))      # (this GPU's joules over this GPU's samples)
    if train:
        out["schedule"] = {"wgrad_side_stream": overlap,
                           "serial_ms_per_step": round(1e3 * serial_el, 3) if serial_el else None,
                           "note": ("timed steps: weight-gradient GEMMs on me_block_bwd's side stream beside the dY -> dX chain; roofline / "
                                    "other_kernels: per-kernel HIP-event durations from a serial pass of the same step (shares are of the "
                                    "serial step)") if overlap else "one stream, program order (timed steps and per-kernel records alike)"}
    if fwd is not None:
        fwd_el, fwd_prof = fwd
        fs = fwd_el / args.steps
        fv = world * B * args.steps / fwd_el
        out["fwd"] = {"what": "encoder forward alone (torch.no_grad), same tokens and weights, same K steps",
                      "ms_per_step": round(1e3 * fs, 3), "samples_per_s": round(fv, 2),
                      "model_tflops_per_s": round(fv * fwd_flops / 1e12, 2),
                      "mfma_frac": round(fv * fwd_flops / 1e12 / ((PEAK_X3_TFLOPS if x3 else PEAK_F32_TFLOPS if f32 else PEAK_BF16_TFLOPS) * world), 4),
                      "roofline": gemm_roofline(fwd_prof, fs, psteps, False),
                      "other_kernels": other_kernels(fwd_prof, fs, psteps)}
        if power_info.get("fwd"):
            out["fwd"]["power"] = power_info["fwd"]
    if not args.no_cpu_baseline and world == 1:
        # the CPU leg runs in its own process (fresh OpenMP pool, hard timeout) so it can never stall the bench
        cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--depth", str(L), "--dim", str(C), "--heads", str(H),
               "--tokens", str(N), "--budget-s", str(args.cpu_budget_s)] + ([] if train else ["--forward-only"])
        try:
            r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=args.cpu_budget_s * 8 + 120)
            cb = json.loads(r.stdout.strip().splitlines()[-1])
            cb["value"] = round(cb["value"], 3)
            out["cpu_baseline"] = cb
        except Exception as e:          # noqa: BLE001 -- a baseline failure must not lose the GPU measurement
            out["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": None, "kind": "port",
                                   "sample": f"failed: {type(e).__name__}: {e}"[:300]}
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
