"""Socket power and clock held over a sustained loop of the bench's forward / train step (dev tool; round 6).

    python tools/power_probe.py [--mode fwd|train] [--secs 3] [--rest-us 0 20 50 100]

Samples the GPU's hwmon power1_average (W) and freq1_input (shader clock) every 10 ms from a host thread while the step loops, after a
0.5 s settle.  --rest-us R inserts R microseconds of idle GPU time (a one-wave s_sleep kernel from torch) behind every Block: if the
step's COMPUTE time shrinks by about what the rests add, the step is bound by the socket's power cap (energy), not by its kernels'
schedules -- rests bank power budget that the next GEMMs spend as clock.  Prints one line per arm:
    mode rest_us ms_per_step ms_compute(= ms_per_step - rests) mean_W max_W mean_MHz J_per_step
Reproduces profiles/r06_power.txt.
"""
import argparse
import glob
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metatransformer_amd as M  # noqa: E402
from metatransformer_amd import parallel  # noqa: E402


class Sampler:
    def __init__(self):
        self.pw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") or glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
        self.fq = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")
        self.w, self.f, self.stop = [], [], False

    @staticmethod
    def _read(paths):
        best = 0.0
        for p in paths:
            try:
                best = max(best, float(open(p).read().strip()) * 1e-6)
            except (OSError, ValueError):
                pass
        return best

    def run(self):
        while not self.stop:
            w, f = self._read(self.pw), self._read(self.fq)
            if w > 0:
                self.w.append(w)
            if f > 0:
                self.f.append(f)
            time.sleep(0.01)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="fwd", choices=["fwd", "train"])
    ap.add_argument("--secs", type=float, default=3.0)
    ap.add_argument("--rest-us", type=float, nargs="*", default=[0.0])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    L, C, H, B, N = 12, 768, 12, 256, 197
    enc = M.build_encoder(L, C, H).to(dev)
    for p in enc.parameters():
        if p.dim() == 2:
            torch.nn.init.normal_(p, std=0.02)
    for b in enc:
        b.compute_dtype = torch.bfloat16
    g = torch.Generator().manual_seed(1000)
    x = torch.randn(B, N, C, generator=g).to(dev).bfloat16()
    gy = (torch.randn(B, N, C, generator=g) / (B * N)).to(dev).bfloat16()
    train = a.mode == "train"
    if train:
        enc.train()
        flat = parallel.FlatParams(enc.named_parameters(), no_decay=parallel.no_decay_rule)
        opt = parallel.FusedAdamW(flat, lr=1e-4, weight_decay=0.05)
        x.requires_grad_(True)
    else:
        enc.eval()

    def rest(us):
        if us > 0:
            torch.cuda._sleep(int(us * 1e-6 * rate))      # (device-side spin of `cycles` clocks: one wave, next to no power)

    # cycles per second of torch.cuda._sleep's counter, calibrated once
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
    rate = 10_000_000 / (e0.elapsed_time(e1) * 1e-3)

    def step(us):
        if train:
            flat.zero_grad()
            x.grad = None
            h = x
            for b in enc:
                h = b(h)
                rest(us)
            h.backward(gy)
            opt.step()
        else:
            with torch.no_grad():
                h = x
                for b in enc:
                    h = b(h)
                    rest(us)

    print(f"# {a.mode}: [256,197,768] bf16, 12 blocks; power sampled every 10 ms over {a.secs} s after 0.5 s; rest = idle GPU time behind every Block")
    print("# mode rest_us  ms_per_step  ms_minus_rests  mean_W  max_W  mean_MHz  J_per_step")
    for us in a.rest_us:
        for _ in range(5):
            step(us)
        torch.cuda.synchronize()
        t_end = time.time() + 0.5
        while time.time() < t_end:
            step(us)
        torch.cuda.synchronize()
        s = Sampler()
        th = threading.Thread(target=s.run)
        th.start()
        n, t0 = 0, time.time()
        e0.record()
        while time.time() - t0 < a.secs:
            for _ in range(10):
                step(us)
            n += 10
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        s.stop = True
        th.join()
        ms = e0.elapsed_time(e1) / n
        w = sum(s.w) / max(1, len(s.w))
        f = sum(s.f) / max(1, len(s.f))
        print(f"{a.mode:5s} {us:7.0f}  {ms:10.3f}  {ms - L * us * 1e-3:13.3f}  {w:7.1f} {max(s.w or [0]):7.1f}  {f:8.0f}  {w * ms * 1e-3:8.2f}", flush=True)


if __name__ == "__main__":
    main()
