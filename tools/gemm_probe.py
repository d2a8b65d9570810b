"""Dev probes of the g256 NT kernel through the raw C ABI: aliasing operand rows (lda/ldb = 0 -> every tile re-reads the
same lines: ~100 % L2/L1 hits) and skipping the epilogue isolate where the time goes.  Run per-env in subprocesses."""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [("qkv_fwd", 256 * 197, 2304, 768), ("proj_fwd", 256 * 197, 768, 768), ("fc2_fwd", 256 * 197, 768, 3072)]


def child():
    import torch
    from metatransformer_amd import _capi
    lib = _capi.load()
    dev = torch.device("cuda:0")
    out = {}
    for name, M, N, K in CASES:
        a = torch.randn(M, K).bfloat16().to(dev); b = (0.05 * torch.randn(N, K)).bfloat16().to(dev)
        c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        for tag, lda, ldb in (("normal", K, K), ("aliasA", 0, K), ("aliasAB", 0, 0)):
            d = _capi.GemmDesc()
            d.op, d.ab_dtype, d.M, d.N, d.K = 0, 1, M, N, K
            d.A, d.lda, d.B, d.ldb = a.data_ptr(), lda, b.data_ptr(), ldb
            d.C, d.ldc, d.c_dtype, d.alpha = c.data_ptr(), N, 1, 1.0
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(3):
                _capi.check(lib.me_gemm(ctypes.byref(d), st))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lib.me_gemm(ctypes.byref(d), st)
            e1.record(); torch.cuda.synchronize()
            us = 100 * e0.elapsed_time(e1)
            out[f"{name}/{tag}"] = round(2.0 * M * N * K / us / 1e6, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    base = {"ME_G2_SCHED": "1", "ME_GEMM_KERNEL": os.environ.get("PROBE_KERNEL", "g2b_256")}
    for env_name, env in (("staggered", {}), ("+static prio trailing", {"ME_G256_DEBUG": "128"}), ("+setprio around mfma", {"ME_G256_DEBUG": "256"}),
                          ("both", {"ME_G256_DEBUG": "384"}), ("staggered again", {})):
        env = dict(base, **env)
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(env_name, line[-1] if line else r.stderr[-800:])
