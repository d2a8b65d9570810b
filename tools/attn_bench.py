"""Time me_attention_fwd / _bwd on encoder shapes (rotating buffers so the 256 MiB Infinity Cache does not flatter it)."""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metatransformer_amd import ops


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="256x197x12x64,128x512x16x64,64x1568x12x64,256x49x12x64")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for spec in args.shapes.split(","):
        B, N, H, hd = map(int, spec.split("x"))
        C = H * hd
        nbuf = 3
        qkvs = [torch.randn(B * N, 3 * C, device=dev).bfloat16() for _ in range(nbuf)]
        dos = [torch.randn(B * N, C, device=dev).bfloat16() for _ in range(nbuf)]
        outs = [ops.attention_fwd(q, B, N, H, hd, hd ** -0.5, True) for q in qkvs]
        i = [0]

        def fwd():
            i[0] = (i[0] + 1) % nbuf
            ops.attention_fwd(qkvs[i[0]], B, N, H, hd, hd ** -0.5, True)

        def bwd():
            i[0] = (i[0] + 1) % nbuf
            o, l = outs[i[0]]
            ops.attention_bwd(qkvs[i[0]], o, dos[i[0]], l, B, N, H, hd, hd ** -0.5)

        tf, tb = timeit(fwd), timeit(bwd)
        fl = 4.0 * B * H * N * N * hd
        io_f = B * N * C * 2 * 4
        io_b = B * N * C * 2 * (3 + 1 + 1 + 3)
        print(f"{spec:18s} fwd {tf:8.1f} us {fl / tf / 1e6:7.1f} TF {io_f / tf / 1e6:6.2f} TB/s | "
              f"bwd {tb:8.1f} us {2.5 * fl / tb / 1e6:7.1f} TF {io_b / tb / 1e6:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
