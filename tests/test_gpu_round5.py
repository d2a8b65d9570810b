"""GPU, round 5: the reference's own batch shapes through the small-M GEMM plans (64 x 128 / 128 x 128 /
128 x 256 tiles chosen by tile count, no K split), the straight-line forms of the "saved gelu'" epilogue pair in the
one-tile-per-workgroup kernels, Block under torch.inference_mode(), and the hipGraph replay of encoder_forward_inference.
Everything goes through the C ABI; the checker is the CPU oracle (oracle/block_oracle.py) or fp64 torch on the same operands."""
import pytest
import torch

from conftest import TOL_BF16_GRAD, TOL_BF16_OP, TOL_BF16_STREAM1, TOL_F32, check_close, rel_err
import metatransformer_amd as M
from metatransformer_amd import _capi, ops
from oracle import block_oracle as bo

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def make_encoder(depth, dim, heads, dev, dtype=torch.float32, seed=3, eps=1e-5):
    from functools import partial
    enc = M.build_encoder(depth, dim, heads, norm_layer=partial(torch.nn.LayerNorm, eps=eps))
    enc.load_state_dict(bo.make_encoder_state_dict(depth, dim, seed=seed), strict=True)
    return enc.to(dev).to(dtype).eval()


# ---- GEMM plans at the reference's row counts: M = B * N of Time-Series (32 x 96), Tabular (256 x 16), Graph (128 x 50),
# X-Ray (32 x 197), PointCloud (32 x 257), one sample (197) and a ragged count, on the encoder's four Linear shapes
SMALL_M = [197, 3072, 4096, 6400, 6304, 8224, 777]
LINEAR_SHAPES = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]


@pytest.mark.parametrize("M_", SMALL_M)
@pytest.mark.parametrize("N,K", LINEAR_SHAPES)
def test_small_m_gemm_plans_every_epilogue(dev, M_, N, K):
    """bias / + GELU / + residual / saved gelu' pair, bf16, per element against fp64 on the same (bf16-exact) operands"""
    dt = torch.bfloat16
    a, w, bias = rnd(M_, K, seed=1).to(dt), (0.05 * rnd(N, K, seed=2)).to(dt), 0.1 * rnd(N, seed=3)
    ad, wd = a.to(dev), w.to(dev)
    lin = a.double() @ w.double().t() + bias.double()
    y = ops.gemm(ad, wd, bias=bias.to(dev))
    check_close(y.float(), lin, TOL_BF16_OP, "bias")
    y = ops.gemm(ad, wd, bias=bias.to(dev), act=_capi.ME_ACT_GELU)
    check_close(y.float(), bo.gelu_erf(lin), TOL_BF16_OP, "gelu")
    res = rnd(M_, N, seed=4).to(dt)
    y = ops.gemm(ad, wd, bias=bias.to(dev), residual=res.to(dev))
    check_close(y.float(), lin + res.double(), TOL_BF16_OP, "residual")
    # ME_GEMM_SAVE_GELU_GRAD (EPI 7 in the one-tile kernels) and ME_GEMM_AUX_IS_FACTOR (EPI 6)
    h = lin.clone().requires_grad_(True)
    bo.gelu_erf(h).sum().backward()
    sav = torch.empty(M_, N, dtype=dt, device=dev)
    y = ops.gemm(ad, wd, bias=bias.to(dev), act=_capi.ME_ACT_GELU, preact=sav, flags=_capi.ME_GEMM_SAVE_GELU_GRAD)
    check_close(y.float(), bo.gelu_erf(lin), TOL_BF16_OP, "gelu (saving gelu')")
    check_close(sav.float(), h.grad, TOL_BF16_OP, "saved gelu'")
    fac = rnd(M_, N, seed=6).to(dt)
    y = ops.gemm(ad, wd, aux=fac.to(dev), flags=_capi.ME_GEMM_AUX_IS_FACTOR, out_dtype=dt)
    check_close(y.float(), (a.double() @ w.double().t()) * fac.double(), TOL_BF16_OP, "x saved factor")


@pytest.mark.parametrize("M_,N,K", [(197, 768, 3072), (394, 768, 2304), (64, 768, 3072), (3072, 768, 9216), (1576, 768, 9216)])
def test_small_m_long_reduction_split(dev, M_, N, K):
    """a handful of 64 x 128 tiles with a long reduction still takes the whole-problem K split (fp32 slabs + deterministic fold); so do
    up to half the chip's slots when K >= 6 144 (the three-plane fc2 of an ME_BF16X3 Block at the reference's batch sizes)"""
    dt = torch.bfloat16
    a, w, bias, res = rnd(M_, K, seed=1).to(dt), (0.05 * rnd(N, K, seed=2)).to(dt), 0.1 * rnd(N, seed=3), rnd(M_, N, seed=4).to(dt)
    y1 = ops.gemm(a.to(dev), w.to(dev), bias=bias.to(dev), residual=res.to(dev))
    y2 = ops.gemm(a.to(dev), w.to(dev), bias=bias.to(dev), residual=res.to(dev))
    assert torch.equal(y1, y2)                       # deterministic
    check_close(y1.float(), a.double() @ w.double().t() + bias.double() + res.double(), TOL_BF16_OP, "split + fold")


# ---- the reference's batches through the encoder (two blocks: the CPU oracle finishes in seconds), forward and dL/dx through
# frozen blocks -- the mode PointCloud / Time-Series / Graph / Hyper-spectral / Tabular drive it in
REF_BATCHES = [("timeseries", 32, 96, 12), ("graph", 128, 50, 32), ("pointcloud", 32, 257, 12), ("hyperspectral", 64, 201, 12),
               ("tabular", 256, 16, 12)]      # (graph, tabular: N <= 64 -> the one-workgroup-per-head attention kernels, fp32 and bf16)


@pytest.mark.parametrize("name,B,N,H", REF_BATCHES)
def test_reference_batch_shapes_forward_and_dx_vs_oracle(dev, name, B, N, H):
    depth, C = 2, 768
    sd = bo.make_encoder_state_dict(depth, C, seed=5)
    g = torch.Generator().manual_seed(77)
    x, go = torch.randn(B, N, C, generator=g), torch.randn(B, N, C, generator=g)
    y_ref, dx_ref, _ = bo.encoder_forward_backward(x, sd, H, go)
    # fp32 (the reference's default arithmetic)
    enc = make_encoder(depth, C, H, dev, seed=5)
    for p in enc.parameters():
        p.requires_grad_(False)
    xr = x.to(dev).requires_grad_(True)
    y = enc(xr)
    (y * go.to(dev)).sum().backward()
    check_close(y, y_ref, TOL_F32, f"{name} fp32 forward")
    check_close(xr.grad, dx_ref, TOL_F32, f"{name} fp32 dL/dx")
    # bf16 compute on a bf16 token stream (the small-M bf16 plans)
    encb = make_encoder(depth, C, H, dev, torch.bfloat16, seed=5)
    for p in encb.parameters():
        p.requires_grad_(False)
    xb = x.to(dev).bfloat16().requires_grad_(True)
    yb = encb(xb)
    (yb.float() * go.to(dev)).sum().backward()
    check_close(yb.float(), y_ref, 1.5 * TOL_BF16_STREAM1, f"{name} bf16 forward")
    check_close(xb.grad.float(), dx_ref, 2 * TOL_BF16_GRAD, f"{name} bf16 dL/dx")
    with torch.no_grad():                            # inference route (folded LayerNorm where the library takes it)
        yi = encb(x.to(dev).bfloat16())
    check_close(yi.float(), y_ref, 1.5 * TOL_BF16_STREAM1, f"{name} bf16 inference")


# ---- Block under torch.inference_mode() (inference tensors have no version counter)
@pytest.mark.parametrize("B", [2, 128])
def test_block_stack_under_inference_mode(dev, B):
    """B = 128 x 197 takes the folded route whose fc2 epilogue emits LayerNorm statistics (the tag on the output tensor used to
    read y._version, which raises for inference tensors).  Inference tensors carry no version counter, so nothing is handed from
    block to block: the result equals the no_grad one WITHOUT the hand-over bit for bit, and the chained one to bf16 noise."""
    enc = make_encoder(3, 768, 12, dev, torch.bfloat16)
    x = rnd(B, 197, 768, seed=9).to(dev).bfloat16()
    with torch.no_grad():
        chained = enc(x)
        for b in enc:
            b.chain_stats = False
        ref = enc(x)
        for b in enc:
            b.chain_stats = True
    with torch.inference_mode():
        y = enc(x)
        y2 = enc(x.clone())                          # an inference tensor as INPUT, too
    assert torch.equal(y, ref) and torch.equal(y2, ref)
    # (two routes to the same statistics: fp32 rounding of (mean, rstd) moves individual bf16 roundings of a three-block bf16 stream)
    check_close(y.float(), chained.float(), 2.5 * TOL_BF16_OP, "epilogue statistics vs a pass over the tokens")
    with torch.inference_mode():                     # weights created under inference mode: their _version is unreadable as well
        enc2 = make_encoder(1, 256, 4, dev, torch.bfloat16)
        z = enc2(rnd(2, 10, 256, seed=1).to(dev).bfloat16())
    assert torch.isfinite(z.float()).all()


# ---- encoder_forward_inference: hipGraph replay for small batches
def test_encoder_forward_inference_graph_replay(dev):
    enc = make_encoder(4, 768, 12, dev, torch.bfloat16)
    for B, N in ((1, 197), (5, 197), (3, 50)):
        xs = [rnd(B, N, 768, seed=s).to(dev).bfloat16() for s in (1, 2, 3)]
        for x in xs:
            eager = M.encoder_forward_inference(enc, x, graph=False)
            replay = M.encoder_forward_inference(enc, x)                  # default: graph for B * N <= GRAPH_MAX_ROWS
            assert torch.equal(eager, replay)
        keep = M.encoder_forward_inference(enc, xs[0])
        snap = keep.clone()
        M.encoder_forward_inference(enc, xs[1])                           # a later call must not overwrite an earlier result
        assert torch.equal(keep, snap)
    # a weight change re-captures (the graph bakes pointers and values of the compute copies in)
    x = rnd(1, 197, 768, seed=5).to(dev).bfloat16()
    before = M.encoder_forward_inference(enc, x)
    with torch.no_grad():
        enc[0].mlp.fc2.bias.add_(0.5)
    after = M.encoder_forward_inference(enc, x)
    assert torch.equal(after, M.encoder_forward_inference(enc, x, graph=False)) and not torch.equal(after, before)
    # fp32 tokens / weights replay too
    enc32 = make_encoder(2, 256, 4, dev)
    x32 = rnd(2, 33, 256, seed=6).to(dev)
    assert torch.equal(M.encoder_forward_inference(enc32, x32), M.encoder_forward_inference(enc32, x32, graph=False))


# ---- fp32-accurate arithmetic on the bf16 matrix pipe (ME_BF16X3; Block.fp32_mode = "3xbf16")
TOL_3X = 1e-4          # stated bound of the mode (north star for fp32: 1e-3); measured ~1e-5 (three bf16 products, lo x lo dropped: 2^-17)


def test_split3_planes_and_gemm_accuracy(dev):
    """me_split3: hi = bf16(x), lo = bf16(x - hi) in the two plane orders, bit for bit; a bf16 GEMM over the split operands
    reproduces the fp64 product of the fp32 operands to ~1e-5 where plain bf16 operands give ~3e-3."""
    x = rnd(130, 256, seed=1).to(dev)
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    a3, w3 = ops.split3(x), ops.split3(x, right_operand=True)
    assert torch.equal(a3, torch.cat([hi, lo, hi], dim=1)) and torch.equal(w3, torch.cat([hi, hi, lo], dim=1))
    M_, N, K = 777, 768, 1024
    a, w, bias = rnd(M_, K, seed=2), 0.05 * rnd(N, K, seed=3), 0.1 * rnd(N, seed=4)
    ref = a.double() @ w.double().t() + bias.double()
    y = ops.gemm(ops.split3(a.to(dev)), ops.split3(w.to(dev), right_operand=True), bias=bias.to(dev), out_dtype=torch.float32)
    check_close(y, ref, TOL_3X, "3xbf16 GEMM")
    assert rel_err(y, ref) < 3e-5
    plain = ops.gemm(a.to(dev).bfloat16(), w.to(dev).bfloat16(), bias=bias.to(dev), out_dtype=torch.float32)
    assert rel_err(plain, ref) > 20 * rel_err(y, ref)            # (what the split buys)


@pytest.mark.parametrize("M_,N,K", [(300, 1024, 256), (50432 // 8, 1024, 256), (256 * 90 + 77, 768, 768)])
def test_gemm_writes_three_plane_output(dev, M_, N, K):
    """c_dtype ME_BF16X3 is reachable through the C ABI only (no torch dtype): the GELU Linear writes [hi | lo | hi] of its fp32 result
    (small-M plan, one-tile g3 kernel, and -- 273 tiles with a long reduction -- the g3 tail split whose fold writes the planes)"""
    import ctypes
    a, w, bias = rnd(M_, K, seed=5), 0.05 * rnd(N, K, seed=6), 0.1 * rnd(N, seed=7)
    a3, w3 = ops.split3(a.to(dev)), ops.split3(w.to(dev), right_operand=True)
    out = torch.zeros(M_, 3 * N, dtype=torch.bfloat16, device=dev)
    b32 = bias.to(dev)
    d = _capi.GemmDesc()
    d.op, d.ab_dtype, d.M, d.N, d.K = _capi.ME_GEMM_NT, _capi.ME_BF16, M_, N, 3 * K
    d.A, d.lda, d.B, d.ldb = a3.data_ptr(), 3 * K, w3.data_ptr(), 3 * K
    d.C, d.ldc, d.c_dtype = out.data_ptr(), 3 * N, _capi.ME_BF16X3
    d.alpha, d.bias, d.act = 1.0, b32.data_ptr(), _capi.ME_ACT_GELU
    lib = _capi.load()
    wsb = lib.me_gemm_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    d.workspace, d.workspace_bytes = ws.data_ptr(), wsb
    _capi.check(lib.me_gemm(ctypes.byref(d), _capi.stream_ptr()), "me_gemm")
    ref = bo.gelu_erf(a.double() @ w.double().t() + bias.double())
    hi, lo, hi2 = out[:, :N], out[:, N:2 * N], out[:, 2 * N:]
    assert torch.equal(hi, hi2)
    check_close(hi.float() + lo.float(), ref, TOL_3X, "three-plane GELU output")
    # lo sits below hi's last bit (|lo| <= half an ulp of hi = 2^-9 |hi|; the epilogue may form v - hi from the unrounded product, so the
    # exact bits of lo are not pinned here -- the sum above is)
    assert bool((lo.float().abs() <= hi.float().abs() * 2.0 ** -8 + 1e-30).all())


@pytest.mark.parametrize("name", ["base_1blk", "base_12blk", "large_2blk"])
def test_forward_3xbf16_matches_reference_golden(dev, name):
    """the reference's fp32 outputs (tests/golden, generated by its own Block code) at 1e-4 from the three-product mode"""
    import json, os
    import numpy as np
    from conftest import GOLDEN
    from oracle.make_golden import _inputs
    z = np.load(os.path.join(GOLDEN, f"encoder_{name}.npz"), allow_pickle=False)
    c = json.loads(str(z["config"]))
    enc = make_encoder(c["depth"], c["dim"], c["heads"], dev, seed=c["seed"], eps=c["eps"])
    for b in enc:
        b.fp32_mode = "3xbf16"
    x, _ = _inputs(c)
    with torch.no_grad():
        y = enc(x.to(dev))
        y1 = M.encoder_forward_inference(enc, x.to(dev), graph=False)
    assert y.dtype == torch.float32 and torch.equal(y, y1)
    check_close(y[:, ::c["tok_stride"]], torch.from_numpy(z["y"]), TOL_3X, name)
    for b in enc:
        assert b.uses_3xbf16(torch.float32, torch.float32)
        b.fp32_mode = "exact"
    with torch.no_grad():
        ye = enc(x.to(dev))
    assert not torch.equal(ye, y) and rel_err(y, ye) < TOL_3X


@pytest.mark.parametrize("B,N,depth", [(3, 70, 2), (24, 197, 1), (40, 16, 1)], ids=["210_tokens", "4728_tokens", "tabular_16_tokens"])
def test_backward_3xbf16_every_gradient_vs_oracle(dev, B, N, depth):
    """forward, dL/dx and all parameter gradients of a block stack in the three-product mode, per element, against the CPU
    oracle (fp32 torch restatement of the reference Block); ragged token counts; then in-place accumulation into a FlatParams buffer.
    From 4 096 tokens on a weight gradient is ONE launch of the split-K wgrad kernel over three plane segments (csrc/gemm3_x3.hip) with
    the bias gradient on it; below, three me_gemm calls."""
    C, H = 768, 12
    sd = bo.make_encoder_state_dict(depth, C, seed=5)
    g = torch.Generator().manual_seed(42)
    x, go = torch.randn(B, N, C, generator=g), torch.randn(B, N, C, generator=g)
    y_ref, dx_ref, dp_ref = bo.encoder_forward_backward(x, sd, H, go)
    enc = make_encoder(depth, C, H, dev, seed=5).train()
    for b in enc:
        b.compute_dtype = "fp32_3xbf16"
    xr = x.to(dev).requires_grad_(True)
    y = enc(xr)
    (y * go.to(dev)).sum().backward()
    check_close(y, y_ref, TOL_3X, "forward")
    check_close(xr.grad, dx_ref, TOL_3X, "dL/dx")
    for k, p in enc.named_parameters():
        check_close(p.grad, dp_ref[k], TOL_3X, f"d{k}")
    first = {k: p.grad.clone() for k, p in enc.named_parameters()}
    # the same gradients accumulated IN PLACE into a flat buffer (me_block_grads.accumulate: beta = 1 on every term): two passes = 2x
    from metatransformer_amd import parallel
    flat = parallel.FlatParams(enc.named_parameters())
    flat.zero_grad()
    for _ in range(2):
        xr.grad = None
        y = enc(xr)
        (y * go.to(dev)).sum().backward()
    for k, p in enc.named_parameters():
        check_close(p.grad, 2 * first[k], 1e-5, f"accumulated d{k}")


def test_3xbf16_falls_back_to_exact_where_it_is_not_built(dev):
    """widths that are not multiples of 256, bf16 tokens and windowed blocks run the exact path (same results as fp32_mode='exact')"""
    enc = make_encoder(1, 192, 3, dev)
    x = rnd(2, 20, 192, seed=3).to(dev)
    with torch.no_grad():
        ye = enc(x)
        enc[0].fp32_mode = "3xbf16"
        y3 = enc(x)
    assert not enc[0].uses_3xbf16(torch.float32, torch.float32) and torch.equal(ye, y3)
    with pytest.raises(_capi.MetaEncError):
        enc[0].fp32_mode = "tf32"
        enc(x)


# ---- a per-LAYER net under the 12-block bf16 stream bound (3e-2 against a measured 1.3-2.1e-2)
@pytest.mark.parametrize("route", ["blocks", "folded_one_call"])
def test_bf16_stream_layer_by_layer_vs_oracle(dev, route):
    """Every block of the 12-block Base encoder against the oracle evaluated on THAT block's own (bf16) input as the GPU produced
    it: the error of one layer cannot hide in, or be excused by, the rounding the stream accumulated before it.  Bound 1e-2 per
    layer (measured one-block error 5.8e-3), where the end-to-end stream bound has to be 3e-2."""
    from oracle.make_golden import ENCODER_CASES, _inputs
    c = ENCODER_CASES["base_12blk"]
    sd = bo.make_encoder_state_dict(c["depth"], c["dim"], seed=c["seed"])
    enc = make_encoder(c["depth"], c["dim"], c["heads"], dev, torch.bfloat16, seed=c["seed"], eps=c["eps"])
    x, _ = _inputs(c)
    x = torch.cat([x, x.flip(0) * 0.5 + 0.1], dim=0) if x.shape[0] == 1 else x       # (two samples with different scales)
    cur = x.to(dev).bfloat16()
    per_block = bo.split_state_dict(sd)
    worst = 0.0
    with torch.no_grad():
        for i, blk in enumerate(enc):
            if route == "folded_one_call":
                blk.fold_norm = "always"             # LayerNorm folded into qkv / fc1 even at this small batch
            nxt = blk(cur) if route == "blocks" else M.encoder_forward_inference(torch.nn.Sequential(blk), cur, graph=False)
            ref = bo.block_forward(cur.float().cpu(), per_block[i], c["heads"], eps=c["eps"])
            check_close(nxt.float(), ref, TOL_BF16_STREAM1, f"layer {i} ({route})")
            worst = max(worst, rel_err(nxt.float(), ref))
            cur = nxt
    assert worst < TOL_BF16_STREAM1


X3_ATTN_SHAPES = [(2, 197, 12, 64), (1, 64, 2, 64), (3, 37, 2, 64), (1, 1, 1, 64), (1, 257, 2, 64), (1, 513, 1, 64), (2, 1568, 2, 64), (64, 65, 3, 64),
                  # (round 6: tiles past N are skipped) every count of valid 16-row tiles in the last 64-row chunk, row blocks whose last waves own nothing
                  (2, 80, 2, 64), (2, 96, 3, 64), (1, 112, 2, 64), (2, 129, 2, 64), (1, 161, 2, 64), (1, 225, 2, 64), (3, 300, 2, 64)]


@pytest.mark.parametrize("B,N,H,hd", X3_ATTN_SHAPES)
def test_attention_fwd_x3_vs_fp64(dev, B, N, H, hd):
    """me_attention_fwd_x3 (three bf16 products per MFMA operand pair, fp32 softmax) against fp64 attention on the same fp32 qkv:
    output and lse at the mode's 1e-4 bound (measured ~1e-5), the ME_BF16X3 planes = the split of the fp32 output bit for bit,
    ragged query blocks / key chunks (N = 1, 37, 65, 197, 257, 513, 1568), and agreement with the exact-fp32 kernel."""
    qkv = rnd(B * N, 3 * H * hd, seed=B * 1000 + N)
    scale = hd ** -0.5
    q, k, v = qkv.double().reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)) * scale
    ref = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B * N, H * hd)
    lse_ref = torch.logsumexp(s, dim=-1)
    out, lse, o3 = ops.attention_fwd_x3(qkv.to(dev), B, N, H, hd, scale, need_lse=True, planes=True)
    check_close(out, ref, TOL_3X, f"x3 attention {B}x{N}x{H}x{hd}")
    assert rel_err(out, ref) < 3e-5
    assert rel_err(lse, lse_ref) < 1e-5
    assert torch.equal(o3, ops.split3(out))
    exact, _ = ops.attention_fwd(qkv.to(dev), B, N, H, hd, scale, False)
    assert rel_err(out, exact) < 3e-5
    # peaked rows (one key far above the rest, in the LAST chunk): the running-max rescale path
    if N >= 130:
        qk = 0.3 * rnd(B * N, 3 * H * hd, seed=7)
        qk[N - 3, H * hd:H * hd + hd] = 6.0 * qk[5, 0:hd] / qk[5, 0:hd].norm() * 8
        q, k, v = qk.double().reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
        ref2 = (torch.softmax((q @ k.transpose(-2, -1)) * scale, dim=-1) @ v).transpose(1, 2).reshape(B * N, H * hd)
        out2, _, _ = ops.attention_fwd_x3(qk.to(dev), B, N, H, hd, scale)
        check_close(out2, ref2, TOL_3X, "x3 attention, peaked rows")


def test_attention_fwd_x3_rejects_other_head_dims(dev):
    with pytest.raises(_capi.MetaEncError):
        ops.attention_fwd_x3(rnd(40, 3 * 2 * 32, seed=1).to(dev), 1, 40, 2, 32, 32 ** -0.5)
    with pytest.raises(_capi.MetaEncError):
        ops.attention_fwd_x3(rnd(40, 3 * 2 * 64, seed=1).to(dev).bfloat16(), 1, 40, 2, 64, 0.125)


@pytest.mark.parametrize("B,N,H,hd", X3_ATTN_SHAPES)
def test_attention_bwd_x3_vs_fp64(dev, B, N, H, hd):
    """me_attention_bwd_x3 (dQ / dK / dV as three bf16 products per operand pair, P recomputed from the forward's lse) against fp64
    autograd of the same attention, per element and each of dQ / dK / dV on its own scale, at the mode's 1e-4 bound; ragged query and
    key blocks (N = 1, 37, 65, 197, 257, 513, 1568); and against the exact-fp32 backward kernel."""
    qkv = rnd(B * N, 3 * H * hd, seed=B * 1000 + N + 1)
    do = rnd(B * N, H * hd, seed=77)
    scale = hd ** -0.5
    qr = qkv.double().requires_grad_(True)
    q, k, v = qr.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax((q @ k.transpose(-2, -1)) * scale, dim=-1) @ v).transpose(1, 2).reshape(B * N, H * hd)
    ref.backward(do.double())
    out, lse, _ = ops.attention_fwd_x3(qkv.to(dev), B, N, H, hd, scale, need_lse=True)
    dqkv, d3 = ops.attention_bwd_x3(qkv.to(dev), out, do.to(dev), lse, B, N, H, hd, scale, planes=True)
    assert torch.equal(d3, ops.split3(dqkv))                  # the planes the qkv GEMMs of an ME_BF16X3 Block read
    C = H * hd
    whole = float(qr.grad.abs().max())
    for j, nm in enumerate(("dQ", "dK", "dV")):
        got, want = dqkv[:, j * C:(j + 1) * C], qr.grad[:, j * C:(j + 1) * C]
        if float(want.abs().max()) < 1e-3 * whole:       # (N = 1: dQ and dK are identically zero)
            assert float((got.cpu().double() - want).abs().max()) <= TOL_3X * whole, nm
            continue
        check_close(got, want, TOL_3X, f"x3 attention backward {nm} {B}x{N}x{H}x{hd}")
    exact = ops.attention_bwd(qkv.to(dev), out, do.to(dev), lse, B, N, H, hd, scale)
    assert rel_err(dqkv, exact) < 5e-5


def _gemm_raw(dev, a3, w3, M_, N, K3, c_dtype, out, ldc, bias=None, act=0, preact=None, aux=None, flags=0):
    """me_gemm through the C ABI with a three-plane output (no torch dtype for it)"""
    import ctypes
    d = _capi.GemmDesc()
    d.op, d.ab_dtype, d.M, d.N, d.K = _capi.ME_GEMM_NT, _capi.ME_BF16, M_, N, K3
    d.A, d.lda, d.B, d.ldb = a3.data_ptr(), K3, w3.data_ptr(), K3
    d.C, d.ldc, d.c_dtype = out.data_ptr(), ldc, c_dtype
    d.alpha, d.act, d.flags = 1.0, act, flags
    if bias is not None:
        d.bias = bias.data_ptr()
    if preact is not None:
        d.preact, d.ldpre, d.preact_dtype = preact.data_ptr(), N, _capi.ME_F32
    if aux is not None:
        d.aux, d.ldaux, d.aux_dtype = aux.data_ptr(), N, _capi.ME_F32
    lib = _capi.load()
    wsb = lib.me_gemm_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    d.workspace, d.workspace_bytes = ws.data_ptr(), wsb
    _capi.check(lib.me_gemm(ctypes.byref(d), _capi.stream_ptr()), "me_gemm")


@pytest.mark.parametrize("M_", [256 * 40 + 100, 256 * 52])
def test_three_plane_mlp_epilogues_on_the_one_tile_kernel(dev, M_):
    """The MLP of an ME_BF16X3 Block at sizes the 256 x 256 one-tile kernel takes (>= 128 tiles): fc1 = bias + erf GELU -> planes, with
    gelu'(h) saved in fp32 (straight-line EPI 9), and the fc2 dgrad = acc x saved fp32 factor -> planes (EPI 10); ragged last tile row."""
    N, K = 1024, 256
    a, w, bias = rnd(M_, K, seed=5), 0.05 * rnd(N, K, seed=6), 0.1 * rnd(N, seed=7)
    a3, w3 = ops.split3(a.to(dev)), ops.split3(w.to(dev), right_operand=True)
    h = (a.double() @ w.double().t() + bias.double()).requires_grad_(True)
    bo.gelu_erf(h).sum().backward()
    out = torch.zeros(M_, 3 * N, dtype=torch.bfloat16, device=dev)
    sav = torch.zeros(M_, N, dtype=torch.float32, device=dev)
    _gemm_raw(dev, a3, w3, M_, N, 3 * K, _capi.ME_BF16X3, out, 3 * N, bias=bias.to(dev), act=_capi.ME_ACT_GELU, preact=sav,
              flags=_capi.ME_GEMM_SAVE_GELU_GRAD)
    hi, lo, hi2 = out[:, :N], out[:, N:2 * N], out[:, 2 * N:]
    assert torch.equal(hi, hi2)
    check_close(hi.float() + lo.float(), bo.gelu_erf(h.detach()), TOL_3X, "fc1: GELU planes")
    check_close(sav, h.grad, TOL_3X, "fc1: saved gelu'")
    out2 = torch.zeros(M_, 3 * N, dtype=torch.bfloat16, device=dev)      # without the saved tensor (inference)
    _gemm_raw(dev, a3, w3, M_, N, 3 * K, _capi.ME_BF16X3, out2, 3 * N, bias=bias.to(dev), act=_capi.ME_ACT_GELU)
    assert torch.equal(out2, out)
    fac = rnd(M_, N, seed=9).to(dev)
    out3 = torch.zeros(M_, 3 * N, dtype=torch.bfloat16, device=dev)
    _gemm_raw(dev, a3, w3, M_, N, 3 * K, _capi.ME_BF16X3, out3, 3 * N, aux=fac, flags=_capi.ME_GEMM_AUX_IS_FACTOR)
    ref = (a.double() @ w.double().t()) * fac.cpu().double()
    check_close(out3[:, :N].float() + out3[:, N:2 * N].float(), ref, TOL_3X, "fc2 dgrad: x saved factor, planes")
    assert torch.equal(out3[:, :N], out3[:, 2 * N:])


@pytest.mark.parametrize("M_,N,K", [(256 * 40 + 100, 1024, 256), (256 * 90 + 77, 768, 768), (256 * 90 + 77, 768, 3072), (50432, 768, 768)])
def test_fp32_residual_epilogue_on_the_one_tile_kernel(dev, M_, N, K):
    """bias + fp32 residual -> fp32 with bf16 operands: proj / fc2 of the reference's AUTOCAST recipes (fp32 tokens, bf16 compute) and
    of the ME_BF16X3 Blocks -- the straight-line EPI 8 of the 256 x 256 one-tile kernel (whole tiles, ragged last row, and the tail
    split whose fold applies the same epilogue), per element against fp64."""
    dt = torch.bfloat16
    a, w, bias, res = rnd(M_, K, seed=1).to(dt), (0.05 * rnd(N, K, seed=2)).to(dt), 0.1 * rnd(N, seed=3), rnd(M_, N, seed=4)
    y = ops.gemm(a.to(dev), w.to(dev), bias=bias.to(dev), residual=res.to(dev), out_dtype=torch.float32)
    ref = a.double() @ w.double().t() + bias.double() + res.double()
    check_close(y, ref, 2e-5, "bias + fp32 residual -> fp32")
    y2 = ops.gemm(a.to(dev), w.to(dev), bias=bias.to(dev), residual=res.to(dev), out_dtype=torch.float32)
    assert torch.equal(y, y2)


@pytest.mark.parametrize("mode", ["bf16_tokens", "autocast_fp32_tokens", "fp32_3xbf16"])
def test_layer_scale_block_at_a_size_the_one_tile_kernel_takes(dev, mode):
    """x + gamma * branch(x) (the detection / segmentation ViT's layer scale, vit.py:313-316) at B x N = 64 x 197: proj / fc2 run on the
    256 x 256 one-tile kernel, whose residual epilogues (EPI 2: bf16 stream, EPI 8: fp32 stream) carry the per-column scale -- against
    the oracle, inference (the forward with gamma inside me_block_fwd)."""
    torch.manual_seed(3)
    blk = M.Block(768, 12, qkv_bias=True, layer_scale=True).to(dev).eval()
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 2:
                torch.nn.init.normal_(p, std=0.02)
        blk.gamma1.uniform_(0.1, 1.5)
        blk.gamma2.uniform_(0.1, 1.5)
    sd = {k: v.detach().cpu().float() for k, v in blk.state_dict().items()}
    x = rnd(64, 197, 768, seed=4)
    ref = bo.block_forward(x, sd, 12, gamma1=sd["gamma1"], gamma2=sd["gamma2"])
    with torch.no_grad():
        if mode == "bf16_tokens":
            blk.compute_dtype = torch.bfloat16
            y = blk(x.to(dev).bfloat16()).float()
            tol = TOL_BF16_STREAM1
        elif mode == "autocast_fp32_tokens":
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = blk(x.to(dev))
            assert y.dtype == torch.float32
            tol = TOL_BF16_STREAM1
        else:
            blk.fp32_mode = "3xbf16"
            y = blk(x.to(dev))
            tol = TOL_3X
    check_close(y, ref, tol, f"layer-scale block, {mode}")


# ---- the patch embed as one kernel (me_patch_embed / me_patch_embed_wgrad, csrc/patch_embed.hip): the gather of Data2Seq/Image.py:19-28
# and Video/models/modeling_finetune.py:283-297 inside the GEMM's operand stager.  Checker: fp64 on the same bf16-exact operands, the
# patches taken by oracle slicing (oracle/tokenizer_oracle.py).
from oracle import tokenizer_oracle as to

PE_CASES = [
    # (x shape, geom (kt, kh, kw, st, sh, sw), Cout, pos, prefix_rows)
    ((3, 3, 224, 224), (1, 16, 16, 1, 16, 16), 768, True, 1),          # Data2Seq.Image + fused pos-embed behind a cls row
    ((5, 3, 240, 208), (1, 16, 16, 1, 16, 16), 768, False, 0),         # 15 x 13 patches: ragged rows, 975 tokens = 3.8 row tiles
    ((2, 3, 4, 64, 64), (2, 16, 16, 2, 16, 16), 768, False, 0),        # tubelets (2, 16, 16): six planes
    ((2, 3, 6, 64, 96), (2, 16, 16, 2, 16, 16), 384, True, 0),         # three tubelet steps, Cout = 1.5 column tiles
    ((4, 1, 64, 128), (1, 8, 32, 1, 8, 32), 256, False, 2),            # kh x kw = 8 x 32: two patch rows per K-tile, one plane
    ((2, 2, 48, 64), (1, 32, 8, 1, 16, 8), 264, False, 0),             # 32 x 8 patches overlapping vertically (stride 16), Cout % 256 != 0
    ((1, 3, 16, 16), (1, 16, 16, 1, 16, 16), 768, False, 0),           # a single token
    ((2, 1, 128, 100), (1, 16, 16, 1, 10, 10), 768, True, 0),          # round 6: the spectrogram tokenizer (Data2Seq/Acoustic.py, ast_models.py:86): stride 10,
                                                                       # overlapping patches whose rows start at byte 20 tx (dword-, not 16-byte-aligned)
    ((3, 1, 128, 256), (1, 16, 16, 1, 10, 10), 768, False, 1),         # 12 x 25 tokens per clip
    ((2, 3, 224, 220), (1, 16, 16, 1, 16, 16), 768, False, 0),         # an image width that is not a multiple of 8: patch rows 8-byte aligned only
]


def _patches(x, geom):
    kt, kh, kw, st, sh, sw = geom
    if x.dim() == 4:
        return to.patchify_2d(x, kh, kw, sh, sw)
    assert (st, sh, sw) == (kt, kh, kw)
    return to.patchify_3d(x, kt, kh, kw)


@pytest.mark.parametrize("case", PE_CASES, ids=lambda c: "x".join(map(str, c[0])) + "_k" + "x".join(map(str, c[1][:3])))
def test_patch_embed_gathers_inside_the_gemm(dev, case):
    shape, geom, Cout, with_pos, prefix = case
    dt = torch.bfloat16
    x = rnd(*shape, seed=11).to(dt)
    cols = _patches(x.float(), geom)                       # [B, tokens, K] fp32, bf16-exact values
    B, tokens, K = cols.shape
    w = (0.05 * rnd(Cout, K, seed=12)).to(dt)
    bias = 0.1 * rnd(Cout, seed=13)
    pos = rnd(tokens, Cout, seed=14) if with_pos else None
    xd, wd = x.to(dev), w.to(dev)
    assert ops.patch_embed_fused(xd, geom, dt, Cout), "this geometry is meant to run fused"
    ref = cols.double() @ w.double().t() + bias.double()
    if pos is not None:
        ref = ref + pos.double()
    # fp32 output: what is left is the accumulation order
    y, tps = ops.patch_embed(xd, wd, bias.to(dev), None if pos is None else pos.to(dev), geom, prefix, torch.float32)
    assert tps == tokens and y.shape == (B * (tokens + prefix), Cout)
    y = y.reshape(B, tokens + prefix, Cout)
    if prefix:
        assert torch.all(y[:, :prefix] == 0)
    check_close(y[:, prefix:], ref, 2e-5, "fused patch embed, fp32 out")
    yb, _ = ops.patch_embed(xd, wd, bias.to(dev), None if pos is None else pos.to(dev), geom, prefix, dt)
    check_close(yb.reshape(B, tokens + prefix, Cout)[:, prefix:].float(), ref, TOL_BF16_OP, "fused patch embed, bf16 out")
    # the two-pass route (me_patchify + me_gemm) on the same operands
    c2, _ = ops.patchify(xd, *geom, dt)
    assert torch.equal(c2.cpu().float().reshape(B, tokens, K), cols)
    y2 = ops.gemm(c2, wd, bias=bias.to(dev), out_dtype=torch.float32).reshape(B, tokens, Cout)
    check_close(y[:, prefix:], y2.double().cpu() + (0 if pos is None else pos.double()), 2e-5, "fused against two-pass")
    # parameter gradients: dW = dY^T patches (fp32 out), db = colsum(dY).  At this size the planner keeps the weight gradient on the
    # small split-K family (two passes inside the entry point); the gathering wgrad kernel takes over from 4 096 tokens on (next test)
    dy = rnd(B * tokens, Cout, seed=15).to(dt)
    dw, db = ops.patch_embed_wgrad(xd, geom, dy.to(dev), torch.float32, True)
    dw_ref = dy.double().t() @ cols.reshape(B * tokens, K).double()
    check_close(dw, dw_ref, 2e-5, "patch-embed wgrad")
    check_close(db, dy.double().sum(0), 2e-5, "bias gradient")
    dw2, db2 = ops.patch_embed_wgrad(xd, geom, dy.to(dev), dt, False)
    assert db2 is None
    check_close(dw2.float(), dw_ref, TOL_BF16_OP, "wgrad, bf16 out")


@pytest.mark.parametrize("case", PE_CASES[1:6] + PE_CASES[7:10], ids=lambda c: "x".join(map(str, c[0][1:])) + "_k" + "x".join(map(str, c[1][:3])))
def test_patch_embed_wgrad_gathers_inside_the_kernel(dev, case):
    """the same geometries with enough samples for the split-K wgrad kernel (>= 4 096 tokens, a ragged count): patches gathered in its B
    stager, patch origins recomputed per K-tile by multiplication; dW and the bias gradient per element against fp64"""
    shape, geom, Cout, _, _ = case
    dt = torch.bfloat16
    one = _patches(torch.zeros(1, *shape[1:]), geom).shape[1]
    B = (4096 + 37 + one - 1) // one
    x = rnd(B, *shape[1:], seed=41).to(dt)
    cols = _patches(x.float(), geom).reshape(B * one, -1)
    dy = rnd(B * one, Cout, seed=42).to(dt)
    xd = x.to(dev)
    assert ops.patch_embed_wgrad_fused(xd, geom, dt, Cout, torch.float32), "meant to run on the gathering wgrad kernel"
    dw, db = ops.patch_embed_wgrad(xd, geom, dy.to(dev), torch.float32, True)
    dw_ref = dy.double().t() @ cols.double()
    check_close(dw, dw_ref, 2e-5, "fused patch-embed wgrad")
    check_close(db, dy.double().sum(0), 2e-5, "bias gradient on the wgrad launch")
    dwb, _ = ops.patch_embed_wgrad(xd, geom, dy.to(dev), dt, False)
    check_close(dwb.float(), dw_ref, TOL_BF16_OP, "fused wgrad, bf16 out")


def test_patch_embed_config2_batch_and_every_token(dev):
    """B = 64 images (12 544 tokens = 49 row tiles x 3 column tiles): every output element against the two-pass route, and the tokenizer
    module end to end (fp32 pixels under bf16 compute take ONE cast pass and the fused kernel) with gradients against nn.Conv2d"""
    dt = torch.bfloat16
    geom = (1, 16, 16, 1, 16, 16)
    x = rnd(64, 3, 224, 224, seed=21).to(dt).to(dev)
    w = (0.03 * rnd(768, 768, seed=22)).to(dt).to(dev)
    bias = (0.1 * rnd(768, seed=23)).to(dev)
    y, _ = ops.patch_embed(x, w, bias, None, geom, 0, dt)
    c2, _ = ops.patchify(x, *geom, dt)
    y2 = ops.gemm(c2, w, bias=bias)
    check_close(y.float(), y2.float(), TOL_BF16_OP, "64 images")
    dy = rnd(64 * 196, 768, seed=24).to(dt).to(dev)
    dw, db = ops.patch_embed_wgrad(x, geom, dy, torch.float32, True)
    dw2 = ops.gemm(dy, c2, op=_capi.ME_GEMM_TN, out_dtype=torch.float32)
    check_close(dw, dw2, 2e-5, "wgrad against the two-pass TN GEMM")
    check_close(db, dy.float().sum(0), 2e-5, "bias gradient")
    # module level
    pe = M.PatchEmbed(img_size=224, patch_size=16, in_c=3, embed_dim=768).to(dev).to(dt)
    conv = torch.nn.Conv2d(3, 768, 16, 16).double()
    conv.load_state_dict({k.replace("proj.", ""): v.detach().cpu().double() for k, v in pe.state_dict().items()})
    xi = rnd(4, 3, 224, 224, seed=25)
    go = rnd(4, 196, 768, seed=26)
    out = pe(xi.to(dev).to(dt))
    (out.float() * go.to(dev)).sum().backward()
    ref = conv(xi.to(dt).double()).flatten(2).transpose(1, 2)
    (ref * go.double()).sum().backward()
    check_close(out.float(), ref, TOL_BF16_OP, "PatchEmbed forward")
    check_close(pe.proj.weight.grad.float(), conv.weight.grad, TOL_BF16_GRAD, "PatchEmbed dW")
    check_close(pe.proj.bias.grad.float(), conv.bias.grad, TOL_BF16_GRAD, "PatchEmbed db")


def test_patch_embed_two_pass_cases_share_the_entry_point(dev):
    """fp32 pixels with fp32 weights (exact arithmetic) are not fusable: same entry points, me_patchify + me_gemm inside.  (The spectrogram's
    stride-10 patches were the other two-pass case until round 6; in bf16 they now run fused, in fp32 they stay here.)"""
    geom = (1, 16, 16, 1, 10, 10)
    x = rnd(2, 1, 128, 100, seed=31)
    cols = to.patchify_2d(x, 16, 16, 10, 10)
    B, tokens, K = cols.shape
    for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, TOL_BF16_OP)):
        xq = x.to(dt)
        w = (0.05 * rnd(256, K, seed=32)).to(dt)
        assert ops.patch_embed_fused(xq.to(dev), geom, dt, 256) == (dt == torch.bfloat16)
        y, tps = ops.patch_embed(xq.to(dev), w.to(dev), None, None, geom, 0, torch.float32)
        ref = to.patchify_2d(xq.float(), 16, 16, 10, 10).double() @ w.double().t()
        check_close(y.reshape(B, tokens, 256), ref, tol, f"two-pass {dt}")
        dy = rnd(B * tokens, 256, seed=33).to(dt)
        dw, db = ops.patch_embed_wgrad(xq.to(dev), geom, dy.to(dev), torch.float32, True)
        check_close(dw, dy.double().t() @ to.patchify_2d(xq.float(), 16, 16, 10, 10).reshape(B * tokens, K).double(), tol, f"two-pass wgrad {dt}")
        check_close(db, dy.double().sum(0), tol, "two-pass bias gradient")


def test_split3_batched_is_split3_per_matrix(dev):
    """me_split3_batched: the planes of each matrix / of its transpose for a batch in one launch, bit for bit what me_split3 (after
    me_transpose_cast) writes one matrix at a time; 50 matrices = two launches; ragged 64 x 64 edge tiles"""
    shapes = [(768, 768), (2304, 768), (768, 3072), (68, 260), (4, 4)] * 10
    ws = [(0.05 * rnd(r, c, seed=100 + i)).to(dev) for i, (r, c) in enumerate(shapes)]
    for transposed in (False, True):
        outs = ops.split3_many(ws, transposed)
        assert len(outs) == len(ws)
        for w, o in zip(ws[:5] + ws[-5:], outs[:5] + outs[-5:]):
            src = ops.transpose_cast(w, torch.float32) if transposed else w
            assert torch.equal(o, ops.split3(src, right_operand=True))
    with pytest.raises(_capi.MetaEncError):
        ops.split3_many([ws[0][:6, :6].contiguous()], False)           # rows / cols not multiples of 4


# ---- seeded random sweeps over the two kernels of this round whose index arithmetic has the most corners
def test_attention_short_sequences_random_sweep(dev):
    """60 random (B, N <= 64, heads, head_dim) problems per dtype through attention_tiny.hip -- every token class (16 / 32 / 64), both
    head-dim classes, head dims that are not tile multiples, odd strides between heads -- forward, lse and all three gradients per
    element against fp64; bf16 head dims are multiples of 8, fp32 of 4 (the kernels' alignment classes)"""
    import random
    rng = random.Random(20260924)
    for dt, tol_o, tol_g in ((torch.bfloat16, 1.5e-2, 3e-2), (torch.float32, 2e-5, 5e-5)):
        step = 8 if dt == torch.bfloat16 else 4
        for case in range(60):
            B, N, H = rng.randint(1, 5), rng.randint(1, 64), rng.randint(1, 6)
            hd = step * rng.randint(1, 64 // step)
            g = torch.Generator().manual_seed(1000 + case)
            qkv = torch.randn(B * N, 3 * H * hd, generator=g).to(dt)
            do = torch.randn(B * N, H * hd, generator=g).to(dt)
            scale = hd ** -0.5
            qr = qkv.double().requires_grad_(True)
            q, k, v = qr.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
            s = (q @ k.transpose(-2, -1)) * scale
            ref = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B * N, H * hd)
            ref.backward(do.double())
            out, lse = ops.attention_fwd(qkv.to(dev), B, N, H, hd, scale, True)
            what = f"{dt} B={B} N={N} H={H} hd={hd}"
            check_close(out.float(), ref.detach(), tol_o, "forward " + what)
            check_close(lse, torch.logsumexp(s.detach(), dim=-1), 5e-3 if dt == torch.bfloat16 else 1e-5, "lse " + what)
            dq_dev = ops.attention_bwd(qkv.to(dev), out, do.to(dev), lse, B, N, H, hd, scale)
            if case % 6 == 0:          # no atomics, fixed summation order: bit-identical on repetition
                out2, lse2 = ops.attention_fwd(qkv.to(dev), B, N, H, hd, scale, True)
                assert torch.equal(out, out2) and torch.equal(lse, lse2), "forward not deterministic, " + what
                assert torch.equal(dq_dev, ops.attention_bwd(qkv.to(dev), out, do.to(dev), lse, B, N, H, hd, scale)), "backward not deterministic, " + what
            dqkv = dq_dev.float().cpu().double()
            C = H * hd
            whole = float(qr.grad.abs().max())
            for name, sl in (("dQ", slice(0, C)), ("dK", slice(C, 2 * C)), ("dV", slice(2 * C, 3 * C))):
                r = qr.grad[:, sl]
                if float(r.abs().max()) < 1e-3 * whole:          # (N = 1: dQ and dK vanish identically)
                    assert float((dqkv[:, sl] - r).abs().max()) <= tol_g * whole, f"{name} {what}"
                else:
                    check_close(dqkv[:, sl], r, tol_g, f"{name} {what}")


def test_patch_embed_random_geometries(dev):
    """40 random fusable geometries (kh x kw = 256 in all four shapes, 1 .. 3 channels, 1 .. 2 frames per tubelet, overlapping and gapped
    strides, ragged token counts, Cout not a multiple of the tile) through the gathering GEMM, and -- with enough samples for the
    split-K wgrad kernel -- through the gathering weight-gradient kernel; fp64 on the same bf16-exact operands"""
    import random
    rng = random.Random(7)
    dt = torch.bfloat16
    fused_wgrads = 0
    for case in range(40):
        kw = rng.choice([8, 16, 32, 64]); kh = 256 // kw
        kt = rng.choice([1, 1, 2]); Cin = rng.randint(1, 3)
        sw = 8 * rng.randint(1, max(1, kw // 8) + 1); sh = rng.randint(max(1, kh // 2), kh + 2); st = kt
        gw, gh, gt = rng.randint(1, 5), rng.randint(1, 5), rng.randint(1, 3)
        W = ((gw - 1) * sw + kw + 7) // 8 * 8; H = (gh - 1) * sh + kh + rng.randint(0, 3); T = gt * kt
        while (H * W) % 8:
            H += 1
        gw, gh = (W - kw) // sw + 1, (H - kh) // sh + 1
        tokens = gt * gh * gw
        big = case % 4 == 0
        B = (4096 + tokens) // tokens + 1 if big else rng.randint(1, 4)
        Cout = 8 * rng.randint(32, 70) if big else 8 * rng.randint(1, 70)
        geom = (kt, kh, kw, st, sh, sw)
        shape = (B, Cin, T, H, W) if kt > 1 or case % 2 else (B, Cin, H, W)
        if len(shape) == 4 and T != 1:
            shape = (B, Cin, T, H, W)
        g = torch.Generator().manual_seed(500 + case)
        x = torch.randn(*shape, generator=g).to(dt)
        x5 = x.float().reshape(B, Cin, T, H, W)
        # the gathered matrix by slicing: token order (t, h, w), feature order (c, dt, dy, dx)
        cols = torch.stack([x5[:, :, pt * st:pt * st + kt, py * sh:py * sh + kh, px * sw:px * sw + kw].reshape(B, -1)
                            for pt in range(gt) for py in range(gh) for px in range(gw)], dim=1)
        K = Cin * kt * 256
        w = (0.05 * torch.randn(Cout, K, generator=g)).to(dt)
        bias = 0.1 * torch.randn(Cout, generator=g)
        xd = x.to(dev)
        what = f"case {case}: x{tuple(shape)} k({kt},{kh},{kw}) s({st},{sh},{sw}) Cout={Cout}"
        assert ops.patch_embed_fused(xd, geom, dt, Cout), what
        y, tps = ops.patch_embed(xd, w.to(dev), bias.to(dev), None, geom, 0, torch.float32)
        assert tps == tokens, what
        check_close(y.reshape(B, tokens, Cout), cols.double() @ w.double().t() + bias.double(), 2e-5, "forward, " + what)
        dy = torch.randn(B * tokens, Cout, generator=g).to(dt)
        fused_wgrads += bool(ops.patch_embed_wgrad_fused(xd, geom, dt, Cout, torch.float32))
        dw, db = ops.patch_embed_wgrad(xd, geom, dy.to(dev), torch.float32, True)
        check_close(dw, dy.double().t() @ cols.reshape(B * tokens, K).double(), 3e-5, "wgrad, " + what)
        check_close(db, dy.double().sum(0), 3e-5, "bias gradient, " + what)
        if case % 5 == 0:
            y2, _ = ops.patch_embed(xd, w.to(dev), bias.to(dev), None, geom, 0, torch.float32)
            dw2, db2 = ops.patch_embed_wgrad(xd, geom, dy.to(dev), torch.float32, True)
            assert torch.equal(y, y2) and torch.equal(dw, dw2) and torch.equal(db, db2), "not deterministic, " + what
    assert fused_wgrads >= 8, f"only {fused_wgrads} cases reached the gathering wgrad kernel"
