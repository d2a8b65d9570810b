"""Rehearsal of the 8-GPU day on ONE GPU: what happens to the training step when a communication kernel
holds R compute units while backward runs.

No multi-GPU box exists in the build loop, and RCCL with world = 1 moves nothing -- but the open design question (DESIGN.md section 6:
the resident NT GEMM and the persistent attention backward CLAIM their work, the weight-gradient kernel does not) only needs the CUs
to be gone.  `tools/cu_hog.hip` holds R CUs (one 160 KiB-LDS workgroup each) for as long as the ring would keep a gradient bucket:
bucket bytes / 90 GB/s per rank (8-rank ring over 153 GB/s xGMI links, 2 (n-1)/n of the bytes each way: 340 MB of fp32 gradients
~ 3.9 ms per step in 6 bucket-sized pieces).  The hog is launched exactly where the all-reduce would be: from the reducer's bucket
hooks, on its own stream behind the kernel that finished the bucket, and the optimizer waits for it (Comm.join semantics).

    python tools/contention.py [--cus 0,8,16,32] [--steps 8] [--gbps 90] [--out profiles/r05_contention.txt]

Per R: ms per step (wall, max of 3 repetitions' min), and the mean HIP-event duration of the NT GEMMs, the weight-gradient GEMMs,
the attention backward and the LayerNorm backward of that step.  R = 0 runs the same code path with a zero-CU hog."""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import metatransformer_amd as M  # noqa: E402
from metatransformer_amd import _capi, ops, parallel  # noqa: E402

HOG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libcuhog.so")


def load_hog():
    lib = ctypes.CDLL(HOG)
    lib.cuhog_launch.restype = ctypes.c_int
    lib.cuhog_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    return lib


class HogComm:
    """parallel.Comm's interface with the all-reduce replaced by `cus` held CUs for bytes / rate seconds (own stream, ordered
    behind the producer stream; join() makes the consumer wait) -- the gradient values are not touched (world = 1 semantics)."""

    def __init__(self, lib, cus: int, gbps: float, device):
        self.lib, self.cus, self.rate = lib, cus, gbps * 1e9
        self.world, self.rank = 8, 0          # (pretend: the reducer is active for world > 1)
        self.stream = torch.cuda.Stream(device=device)
        self.launched = 0
        self.held_us = 0.0

    def allreduce(self, buf: torch.Tensor) -> None:
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        us = buf.numel() * buf.element_size() / self.rate * 1e6
        rc = self.lib.cuhog_launch(self.cus, us, self.stream.cuda_stream)
        if rc:
            raise RuntimeError(f"cuhog_launch failed ({rc})")
        self.launched += 1
        self.held_us += us

    def join(self) -> None:
        torch.cuda.current_stream().wait_stream(self.stream)

    def info(self):
        return {"world": self.world, "rank": self.rank, "buckets_reduced": self.launched}


def build_step(dev, comm, B=256, N=197, L=12, C=768, H=12, wire=None):
    torch.manual_seed(0)
    enc = M.build_encoder(L, C, H).to(dev)
    for p in enc.parameters():
        if p.dim() == 2:
            torch.nn.init.normal_(p, std=0.02)
    for blk in enc:
        blk.compute_dtype = torch.bfloat16
    g = torch.Generator().manual_seed(1000)
    x = torch.randn(B, N, C, generator=g).to(dev).bfloat16().requires_grad_(True)
    gy = (torch.randn(B, N, C, generator=g) / (B * N)).to(dev).bfloat16()
    enc.train()
    flat = parallel.FlatParams(enc.named_parameters(), no_decay=parallel.no_decay_rule)
    opt = parallel.FusedAdamW(flat, lr=1e-4, weight_decay=0.05)
    red = parallel.OverlappedGradReducer(flat, comm=comm, force=True, wire_dtype=wire)

    def step():
        flat.zero_grad()
        x.grad = None
        enc(x).backward(gy)
        red.finish()
        opt.step(grad_scale=1.0 / 8)
    return step, flat


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cus", default="0,8,16,32")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--gbps", type=float, default=90.0)
    ap.add_argument("--out", default="")
    ap.add_argument("--reserve", default="0", help="me_gemm_reserve_cus per run: a number, or 'R' = the CUs held (what me_comm_init does with 16)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = load_hog()
    lines = [f"# tools/contention.py: Base [256,197,768] bf16 train step (fwd + bwd + AdamW), a CU hog in place of the 6 bucket all-reduces "
             f"(bucket bytes / {a.gbps:.0f} GB/s each, launched from the reducer's bucket hooks on its own stream; optimizer joins)",
             f"# device: {torch.cuda.get_device_name(0)};  {a.steps} steps per measurement, best of 3;  per-kernel columns: mean HIP-event "
             f"duration (us) inside that (overlapped) step",
             f"# me_gemm_reserve_cus = {a.reserve} ('R': the CUs held; 0: round 5's one-item-per-CU weight-gradient grid)",
             f"{'R (CUs held)':>12} {'ms/step':>9} {'vs R=0':>8} {'held ms/step':>13} {'NT gemm':>9} {'wgrad':>9} {'attn bwd':>9} {'LN bwd':>8}"]
    base = None
    for R in [int(v) for v in a.cus.split(",")]:
        res = R if a.reserve == "R" else int(a.reserve)
        _capi.load().me_gemm_reserve_cus(res)       # (round 6: weight gradients planned as 256 - res balanced static parts)
        comm = HogComm(lib, R, a.gbps, dev)
        step, _ = build_step(dev, comm)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / a.steps)
        n0, h0 = comm.launched, comm.held_us
        ops.gemm_profile(True)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        recs = ops.gemm_profile_read()
        ops.gemm_profile(False)
        held = (comm.held_us - h0) / 3 / 1e3

        def avg(code, dt=None):
            v = [ms for (op, d, m, n, k, ms) in recs if op == code and (dt is None or d == dt)]
            return 1e3 * sum(v) / len(v) if v else float("nan")
        if base is None:
            base = best
        lines.append(f"{R:>12d} {1e3 * best:>9.3f} {100 * (best / base - 1):>+7.1f}% {held:>13.2f} {avg(_capi.ME_GEMM_NT, _capi.ME_BF16):>9.1f} "
                     f"{avg(_capi.ME_GEMM_TN, _capi.ME_BF16):>9.1f} {avg(_capi.ME_PROF_ATTN_BWD):>9.1f} {avg(_capi.ME_PROF_LN_BWD):>8.1f}")
        print(lines[-1], flush=True)
        del step, comm
        torch.cuda.empty_cache()
    _capi.load().me_gemm_reserve_cus(0)
    text = "\n".join(lines) + "\n"
    if a.out:
        with open(a.out, "w") as f:
            f.write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
