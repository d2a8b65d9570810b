"""GPU, round 6: (1) the folded-LayerNorm GEMMs form their row pairs from the residual launches' 64-column partials themselves
(me_gemm_desc.row_parts: no me_row_stats_combine launch between the GEMMs of a folded forward); (2) the three-product fp32 mode
(ME_BF16X3, Block.fp32_mode = "3xbf16") at the sizes it is benchmarked at -- BASELINE config 2's full batch, a Large two-block
slice, the K = 9 216 reduction at 50 432 rows.  Everything goes through the C ABI; the checker is the CPU
oracle (oracle/block_oracle.py) or fp64 torch on the same operands."""
import ctypes
import os

import pytest
import torch

from conftest import TOL_BF16_OP, TOL_BF16_STREAM12, check_close, rel_err
import metatransformer_amd as M
from metatransformer_amd import _capi, ops
from oracle import block_oracle as bo

pytestmark = pytest.mark.gpu

TOL_3X = 1e-4          # stated bound of the three-product mode (north star for fp32: 1e-3); measured ~1e-5


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def make_encoder(depth, dim, heads, dev, dtype=torch.float32, seed=3, eps=1e-5):
    from functools import partial
    enc = M.build_encoder(depth, dim, heads, norm_layer=partial(torch.nn.LayerNorm, eps=eps))
    enc.load_state_dict(bo.make_encoder_state_dict(depth, dim, seed=seed), strict=True)
    return enc.to(dev).to(dtype).eval()


# ------------------------------------------------------------------------------------------------ (1) row_parts
@pytest.mark.parametrize("M_rows,N_out,K,act,offset", [
    (22016 + 37, 2304, 768, False, 0.0),      # 87 x 9 tiles: whole tiles + a last round of 128-row items, ragged last row tile
    (22016 + 37, 3072, 768, True, 0.0),       # fc1: GELU form
    (22016 + 37, 2304, 768, False, 40.0),     # rows whose mean is tens of times their spread
    (50432, 2304, 768, False, 3.0),           # config 2's qkv launch
    (16384 + 128, 4096, 1024, True, 0.0),     # Large: 4 partials per row
    (65536 + 1, 1536, 512, False, 0.0),       # 2 partials per row, an odd last row
])
def test_folded_gemm_forms_pairs_from_partials(dev, M_rows, N_out, K, act, offset):
    """me_gemm(row_parts) == me_gemm(row_affine = me_row_stats_combine(partials)): the same folded Linear(LayerNorm(x)) [+ GELU], the
    pairs formed in the GEMM's own epilogue.  x is itself the output of a residual launch that left the partials (as in a Block)."""
    eps = 1e-6
    g = torch.Generator().manual_seed(M_rows + N_out + int(offset))
    a0 = torch.randn(M_rows, K, generator=g).bfloat16().to(dev)
    w0 = (torch.randn(K, K, generator=g) * K ** -0.5).bfloat16().to(dev)
    res = (torch.randn(M_rows, K, generator=g) * (0.5 + torch.rand(M_rows, 1, generator=g)) + offset * torch.randn(M_rows, 1, generator=g)).bfloat16().to(dev)
    x, part = ops.gemm(a0, w0, residual=res, want_row_stats=True)
    assert part is not None and part.shape == (K // 256, M_rows, 2)
    gamma, beta = 1.0 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    w = torch.randn(N_out, K, generator=g) * K ** -0.5
    b = 0.1 * torch.randn(N_out, generator=g)
    wf = (w * gamma).bfloat16()
    s = wf.float().sum(1)
    c = w @ beta + b
    wfd, sd_, cd = wf.to(dev), s.to(dev), c.to(dev)
    kw = dict(bias=cd, col_shift=sd_, act=_capi.ME_ACT_GELU if act else _capi.ME_ACT_NONE)
    pairs = ops.row_stats_combine(part, eps)
    y_pairs = ops.gemm(x, wfd, row_affine=pairs, **kw)
    y_parts = ops.gemm(x, wfd, row_parts=part, row_eps=eps, **kw)
    # the two routes differ only in the summation order of the part means and in rsq / rcp against rsqrtf / division: a last-bit difference
    # in rstd / mean moves an output across a bf16 rounding boundary now and then, never further than that rounding step
    diff = (y_parts.float() - y_pairs.float()).abs()
    frac = float((diff > 0).float().mean())
    print(f"row_parts vs row_affine: {frac:.2e} of the outputs differ, max {float(diff.max()):.3g}")
    assert frac < 5e-3
    # (rows far off centre: the output is rstd (acc - mean s) + c, a last-bit step of a mean of ~|offset| x 3 moves EVERY output of the row
    #  by ~1e-5 x s / std whatever its own size)
    assert bool((diff <= 2.0 ** -6 * torch.maximum(y_pairs.float().abs(), y_parts.float().abs()) + 1e-5 + 2e-5 * offset).all())
    # ... and against fp64 LayerNorm -> Linear on a sample of rows (first / last tile, the ragged edge)
    rows = torch.cat([torch.arange(0, 300), torch.arange(M_rows // 2, M_rows // 2 + 300), torch.arange(M_rows - 300, M_rows)])
    xs = x[rows.to(dev)].double().cpu()
    xn = (xs - xs.mean(1, keepdim=True)) / (xs.var(1, unbiased=False, keepdim=True) + eps).sqrt()
    ref = (xn * gamma.double() + beta.double()) @ w.double().t() + b.double()
    if act:
        ref = bo.gelu_erf(ref)
    tol = TOL_BF16_OP * (1.0 + 0.25 * offset)       # (x instead of LN(x) is what gets rounded to bf16: DESIGN section 4, folded LayerNorm)
    check_close(y_parts[rows.to(dev)].float(), ref, tol, "row_parts vs fp64 LayerNorm + Linear")
    # the run is deterministic
    assert torch.equal(y_parts, ops.gemm(x, wfd, row_parts=part, row_eps=eps, **kw))


def test_row_parts_is_refused_where_the_resident_kernel_does_not_run(dev):
    """small problems (fewer tiles than CUs), K not a multiple of 256 or above 1 024: me_gemm_takes_row_parts says no and me_gemm rejects the descriptor
    instead of reading partials as pairs"""
    lib = _capi.load()
    for M_rows, N_out, K in ((4096, 2304, 768), (50432, 2304, 384), (50432, 2304, 1280)):
        x = rnd(M_rows, K, seed=1).bfloat16().to(dev)
        w = rnd(N_out, K, seed=2).bfloat16().to(dev)
        part = torch.zeros(max(K // 256, 1), M_rows, 2, device=dev)
        with pytest.raises(_capi.MetaEncError):
            ops.gemm(x, w, row_parts=part, row_eps=1e-6, col_shift=torch.zeros(N_out, device=dev))
        d = _capi.GemmDesc()
        d.op, d.ab_dtype, d.M, d.N, d.K = _capi.ME_GEMM_NT, _capi.ME_BF16, M_rows, N_out, K
        d.A, d.lda, d.B, d.ldb = x.data_ptr(), K, w.data_ptr(), K
        out = torch.empty(M_rows, N_out, dtype=torch.bfloat16, device=dev)
        d.C, d.ldc, d.c_dtype, d.alpha = out.data_ptr(), N_out, _capi.ME_BF16, 1.0
        cs = torch.zeros(N_out, device=dev)
        d.col_shift, d.row_parts, d.row_nparts, d.row_eps = cs.data_ptr(), part.data_ptr(), max(K // 256, 1), 1e-6
        assert lib.me_gemm_takes_row_parts(ctypes.byref(d)) == 0


def test_folded_forward_runs_without_combine_launches(dev):
    """Block by Block (tensor tags), me_encoder_fwd and the un-chained route agree; the tag now carries the fc2 partials themselves and
    folds to the pairs of the stored output"""
    c = dict(depth=3, dim=768, heads=12, eps=1e-6, seed=41)
    enc = make_encoder(c["depth"], c["dim"], c["heads"], dev, torch.bfloat16, seed=c["seed"], eps=c["eps"])
    B, N = 112, 197
    x = rnd(B, N, 768, seed=9).bfloat16().to(dev)
    with torch.no_grad():
        y = enc(x)
        tag = getattr(y, "_me_ln_stats", None)
        assert tag is not None and tag[0].shape == (3, B * N, 2)
        want = ops.row_stats(y.reshape(B * N, 768), 1e-6)
        got = ops.row_stats_combine(tag[0], 1e-6)
        assert rel_err(got[:, 0], want[:, 0]) < 2e-5
        assert torch.equal(M.encoder_forward_inference(enc, x), y)
        lib = _capi.load()
        lib.me_gemm_profile_enable(1)
        enc(x)
        recs = (_capi.GemmProfileRec * 256)()
        n = lib.me_gemm_profile_read(recs, 256)
        lib.me_gemm_profile_enable(0)
        # one me_row_stats pass for the first block's norm1 (K = 0), no combine launch (K = 1) between the GEMMs
        st = [r.K for r in recs[:n] if r.op == _capi.ME_PROF_ROW_STATS]
        assert st == [0], st


# ------------------------------------------------------------------------------------------------ (2) 3xbf16 at full size
@pytest.mark.slow
def test_3xbf16_config2_full_batch_backward_vs_oracle(dev):
    """BASELINE config 2 at FULL size in the reference's default arithmetic (fp32 tokens, fp32 weights) through the three-product
    mode: y, dL/dx of every sample and all 144 parameter gradients per element at 1e-4 against the CPU oracle's autograd (chunks of
    32 samples, parameter gradients summed).  Exercises what the bench runs: the one-launch three-segment weight gradient over
    3 x 50 432 rows (csrc/gemm3_x3.hip), the K = 9 216 fc2 / fc1-dgrad launches, the three-product attention, in-place accumulation
    into the flat gradient buffer."""
    from metatransformer_amd import parallel
    L, C, Hh, B, N = 12, 768, 12, 256, 197
    sd = bo.make_encoder_state_dict(L, C, seed=91)
    g = torch.Generator().manual_seed(92)
    x = torch.randn(B, N, C, generator=g)
    go = torch.randn(B, N, C, generator=g) / (B * N) ** 0.5
    enc = M.build_encoder(L, C, Hh)
    enc.load_state_dict(sd, strict=True)
    enc = enc.to(dev).train()
    assert M.set_fp32_mode(enc, "3xbf16") == L
    flat = parallel.FlatParams(enc.named_parameters(), no_decay=parallel.no_decay_rule)
    xd = x.to(dev).requires_grad_(True)
    flat.zero_grad()
    y = enc(xd)
    assert all(b.uses_3xbf16(torch.float32, torch.float32) for b in enc)
    y.backward(go.to(dev))
    torch.cuda.synchronize()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    dx_ref, y_ref = torch.empty_like(x), torch.empty_like(x)
    torch.set_num_threads(max(1, min(64, len(os.sched_getaffinity(0)))))
    for i in range(0, B, 32):
        xs = x[i:i + 32].clone().requires_grad_(True)
        ys = bo.encoder_forward(xs, params, Hh)
        (ys * go[i:i + 32]).sum().backward()
        dx_ref[i:i + 32] = xs.grad
        y_ref[i:i + 32] = ys.detach()
    check_close(y, y_ref, TOL_3X, "3xbf16 config 2 y")
    check_close(xd.grad, dx_ref, TOL_3X, "3xbf16 config 2 dL/dx")
    worst = {}
    for k, p in enc.named_parameters():
        kk = k.split(".", 1)[1]
        worst[kk] = max(worst.get(kk, 0.0), rel_err(p.grad, params[k].grad))
        check_close(p.grad, params[k].grad, TOL_3X, f"3xbf16 config 2 d{k}")
    print("3xbf16 config-2 full-batch gradient errors (worst over layers):", {k: f"{v:.1e}" for k, v in worst.items()})


def test_3xbf16_large_two_block_backward_vs_oracle(dev):
    """Large (C = 1024, hidden 4096, 16 heads), 16 x 512 tokens = 8 192 rows (one-launch plane weight gradients, N = 512 three-product
    attention): forward, dL/dx, every parameter gradient per element at 1e-4"""
    sd = bo.make_encoder_state_dict(2, 1024, seed=31)
    g = torch.Generator().manual_seed(33)
    B, N = 16, 512
    x, go = torch.randn(B, N, 1024, generator=g), torch.randn(B, N, 1024, generator=g) / (B * N) ** 0.5
    torch.set_num_threads(max(1, min(64, len(os.sched_getaffinity(0)))))
    y_ref, dx_ref, dp_ref = bo.encoder_forward_backward(x, sd, 16, go)
    enc = M.build_encoder(2, 1024, 16)
    enc.load_state_dict(sd, strict=True)
    enc = enc.to(dev).train()
    M.set_fp32_mode(enc, "3xbf16")
    xr = x.to(dev).requires_grad_(True)
    y = enc(xr)
    assert all(b.uses_3xbf16(torch.float32, torch.float32) for b in enc)
    (y * go.to(dev)).sum().backward()
    check_close(y, y_ref, TOL_3X, "Large 3xbf16 y")
    check_close(xr.grad, dx_ref, TOL_3X, "Large 3xbf16 dL/dx")
    for k, p in enc.named_parameters():
        check_close(p.grad, dp_ref[k], TOL_3X, f"Large 3xbf16 d{k}")


@pytest.mark.parametrize("M_rows,N_out,K", [(50432, 768, 3072), (50432, 3072, 768), (50432, 768, 2304)])
def test_3xbf16_gemm_at_config2_rows(dev, M_rows, N_out, K):
    """the three-plane NT launches of a config-2 step as plain GEMMs (K = 9 216: fc2 / the fc1 dgrad; 6 912: the qkv dgrad), every
    output against an fp64 product of the fp32 operands, and the TN form (one weight gradient over 50 432 rows) against fp64"""
    a, w, bias = rnd(M_rows, K, seed=11), 0.05 * rnd(N_out, K, seed=12), 0.1 * rnd(N_out, seed=13)
    ad, wd = a.to(dev), w.to(dev)
    y = ops.gemm(ops.split3(ad), ops.split3(wd, right_operand=True), bias=bias.to(dev), out_dtype=torch.float32)
    ref = ad.double() @ wd.double().t() + bias.to(dev).double()
    check_close(y, ref, TOL_3X, "3xbf16 NT at 50 432 rows")
    assert rel_err(y, ref) < 3e-5


# ------------------------------------------------------------------------------------------------ ADVICE r5
def test_overlapped_optimizer_refuses_a_block_used_twice(dev):
    """FusedAdamW(overlap=True) updates a Block's parameters as soon as backward has written their gradients; a Block that runs twice in
    one backward (shared encoder, tied weights) would get its second gradient after that update -- it must raise, not train on"""
    from metatransformer_amd import parallel
    enc = make_encoder(1, 256, 4, dev).train()
    for b in enc:
        b.compute_dtype = torch.bfloat16
    flat = parallel.FlatParams(enc.named_parameters(), no_decay=parallel.no_decay_rule)
    opt = parallel.FusedAdamW(flat, lr=1e-3, overlap=True)
    x = rnd(4, 64, 256, seed=2).to(dev).bfloat16().requires_grad_(True)
    flat.zero_grad()
    y = enc(enc(x))                                   # the same Block twice
    with pytest.raises(_capi.MetaEncError, match="second gradient"):
        y.sum().backward()
    del opt


def test_pack_encoder_warms_the_three_product_copies(dev):
    """compute_dtype = "fp32_3xbf16" is a string, not a torch dtype: pack_encoder(warm=True) must build the split three-plane weight
    copies that mode reads (it used to hand the string to the bf16 cast)"""
    from metatransformer_amd import heads
    enc = make_encoder(2, 256, 4, dev)
    for b in enc:
        b.compute_dtype = "fp32_3xbf16"
    flat = heads.pack_encoder(enc, warm=True)
    assert flat.numel > 0
    for b in enc:
        assert all((n, t, dev) in b._wcache._x3 or (n, t, torch.device("cuda", 0)) in b._wcache._x3 for n in ("qkv", "proj", "fc1", "fc2") for t in (False, True))
    x = rnd(2, 70, 256, seed=3).to(dev)
    with torch.no_grad():
        y = enc(x)
    sd = {k: v.detach().cpu() for k, v in enc.state_dict().items()}
    check_close(y, bo.encoder_forward(x.cpu(), sd, 4), TOL_3X, "packed 3xbf16 encoder")


# ------------------------------------------------------------------------------------------------ weight gradients as balanced static parts
@pytest.mark.parametrize("Mo,No,K,reserve", [(2304, 768, 50432, 16), (768, 3072, 50432, 16), (3072, 768, 50432, 8), (768, 768, 50432, 32),
                                             (768, 768, 33 * 197, 16), (1024, 4096, 65536, 16), (768, 2304, 50432, 64)])
def test_wgrad_balanced_partition_vs_fp64(dev, Mo, No, K, reserve):
    """me_gemm_reserve_cus(R): ME_GEMM_TN problems of the 256 x 256 family run on 256 - R workgroups as whole split levels + a shared leftover
    (gemm_g3tn_sk_kernel: a leftover workgroup's K range may end one tile and begin the next; a tile's slabs are folded in K order).  dW and the fused bias gradient against
    fp64 on the same bf16 operands, beta = 0 and 1, deterministic, and the plan is the one reported (me_gemm_profile_rec.plan bit 5)."""
    lib = _capi.load()
    g = torch.Generator().manual_seed(Mo + No + reserve)
    dy = torch.randn(K, Mo, generator=g).bfloat16().to(dev)
    x = torch.randn(K, No, generator=g).bfloat16().to(dev)
    ref = dy.double().t() @ x.double()
    ref_b = dy.double().sum(0)
    prev = lib.me_gemm_reserve_cus(reserve)
    try:
        lib.me_gemm_profile_enable(1)
        dw, db = ops.gemm(dy, x, op=_capi.ME_GEMM_TN, out_dtype=torch.float32, want_colsum_a=True)
        recs = (_capi.GemmProfileRec * 8)()
        n = lib.me_gemm_profile_read(recs, 8)
        lib.me_gemm_profile_enable(0)
        T, slots, upt = ((Mo + 255) // 256) * ((No + 255) // 256), 256 - reserve, ((K + 63) // 64 + 1) // 2
        want_sk = slots % T != 0 and T * upt >= 2 * slots        # (else the uniform split fills the slots / the reduction is too short to cut)
        assert n >= 1 and (recs[0].plan & 15) == 4 and bool(recs[0].plan & 32) == want_sk, hex(recs[0].plan)
        dw2, db2 = ops.gemm(dy, x, op=_capi.ME_GEMM_TN, out_dtype=torch.float32, want_colsum_a=True)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)
        acc, accb = dw.clone(), db.clone()
        ops.gemm(dy, x, op=_capi.ME_GEMM_TN, out=acc, beta=1.0, want_colsum_a=True, colsum_out=accb)
    finally:
        lib.me_gemm_reserve_cus(prev)
    check_close(dw, ref, 2e-5, "dW, balanced parts")
    check_close(db, ref_b, 2e-5, "db, balanced parts")
    check_close(acc, 2 * ref, 2e-5, "dW accumulated")
    check_close(accb, 2 * ref_b, 2e-5, "db accumulated")
    dw0, db0 = ops.gemm(dy, x, op=_capi.ME_GEMM_TN, out_dtype=torch.float32, want_colsum_a=True)      # the one-item-per-CU plan
    assert rel_err(dw, dw0) < 1e-5 and rel_err(db, db0) < 1e-5



# ------------------------------------------------------------------------------------------------ ME_BF16X2: two planes + a wrapped A operand
@pytest.mark.parametrize("M_rows,N_out,Kc,res", [(50432, 768, 3072, True), (50432, 768, 3072, False), (8192 + 77, 1024, 4096, True), (33000, 768, 768, False)])
def test_gemm_reads_two_plane_operand_with_wrapped_reduction(dev, M_rows, N_out, Kc, res):
    """me_gemm_desc.a_wrap_k: A = [hi | lo] planes of an fp32 matrix (lda = 2 Kc), weights ME_BF16X3 [hi | hi | lo], K = 3 Kc -- the kernel re-reads
    the hi plane for the third segment.  Equals the three-plane form bit for bit (same products in the same order) and fp64 at 1e-4."""
    lib = _capi.load()
    a, w, bias = rnd(M_rows, Kc, seed=21).to(dev), (0.05 * rnd(N_out, Kc, seed=22)).to(dev), (0.1 * rnd(N_out, seed=23)).to(dev)
    r = rnd(M_rows, N_out, seed=24).to(dev) if res else None
    a3, w3 = ops.split3(a), ops.split3(w, right_operand=True)
    a2 = a3[:, :2 * Kc].contiguous()                                  # [hi | lo]
    y3 = ops.gemm(a3, w3, bias=bias, residual=r, out_dtype=torch.float32)
    d = _capi.GemmDesc()
    d.op, d.ab_dtype, d.M, d.N, d.K = _capi.ME_GEMM_NT, _capi.ME_BF16, M_rows, N_out, 3 * Kc
    d.A, d.lda, d.B, d.ldb = a2.data_ptr(), 2 * Kc, w3.data_ptr(), 3 * Kc
    y2 = torch.empty(M_rows, N_out, dtype=torch.float32, device=dev)
    d.C, d.ldc, d.c_dtype, d.alpha = y2.data_ptr(), N_out, _capi.ME_F32, 1.0
    d.bias, d.a_wrap_k = bias.data_ptr(), 2 * Kc
    if res:
        d.residual, d.ldres, d.res_dtype = r.data_ptr(), N_out, _capi.ME_F32
    assert lib.me_gemm_takes_a_wrap(ctypes.byref(d)) == 1
    wsb = lib.me_gemm_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    d.workspace, d.workspace_bytes = ws.data_ptr(), wsb
    _capi.check(lib.me_gemm(ctypes.byref(d), _capi.stream_ptr()), "me_gemm")
    assert torch.equal(y2, y3)
    ref = a.double() @ w.double().t() + bias.double() + (r.double() if res else 0)
    check_close(y2, ref, TOL_3X, "two-plane A, wrapped reduction")
    # small problems (another kernel family) say no instead of mis-reading the operand
    d.M = 512
    assert lib.me_gemm_takes_a_wrap(ctypes.byref(d)) == 0 and lib.me_gemm(ctypes.byref(d), _capi.stream_ptr()) != 0


def test_gemm_writes_two_plane_output(dev):
    """c_dtype ME_BF16X2: the MLP epilogues of an ME_BF16X3 Block write [hi | lo] only (fc1: bias + erf GELU with gelu' saved; fc2 dgrad: x the saved
    factor) -- the same hi / lo values as the three-plane form"""
    lib = _capi.load()
    M_rows, N_out, Kc = 50432 // 4, 3072, 768
    a, w, bias = rnd(M_rows, Kc, seed=31).to(dev), (0.05 * rnd(N_out, Kc, seed=32)).to(dev), (0.1 * rnd(N_out, seed=33)).to(dev)
    a3, w3 = ops.split3(a), ops.split3(w, right_operand=True)
    outs = {}
    for cdt, planes in ((_capi.ME_BF16X3, 3), (_capi.ME_BF16X2, 2)):
        out = torch.zeros(M_rows, planes * N_out, dtype=torch.bfloat16, device=dev)
        pre = torch.zeros(M_rows, N_out, dtype=torch.float32, device=dev)
        d = _capi.GemmDesc()
        d.op, d.ab_dtype, d.M, d.N, d.K = _capi.ME_GEMM_NT, _capi.ME_BF16, M_rows, N_out, 3 * Kc
        d.A, d.lda, d.B, d.ldb = a3.data_ptr(), 3 * Kc, w3.data_ptr(), 3 * Kc
        d.C, d.ldc, d.c_dtype, d.alpha = out.data_ptr(), planes * N_out, cdt, 1.0
        d.bias, d.act = bias.data_ptr(), _capi.ME_ACT_GELU
        d.preact, d.ldpre, d.preact_dtype, d.flags = pre.data_ptr(), N_out, _capi.ME_F32, _capi.ME_GEMM_SAVE_GELU_GRAD
        wsb = lib.me_gemm_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
        d.workspace, d.workspace_bytes = ws.data_ptr(), wsb
        _capi.check(lib.me_gemm(ctypes.byref(d), _capi.stream_ptr()), "me_gemm")
        outs[planes] = (out, pre)
    o3, p3 = outs[3]
    o2, p2 = outs[2]
    assert torch.equal(o2, o3[:, :2 * N_out]) and torch.equal(p2, p3)
    ref = bo.gelu_erf(a.double() @ w.double().t() + bias.double())
    check_close(o2[:, :N_out].float() + o2[:, N_out:].float(), ref, TOL_3X, "two-plane GELU output")


# ----------------------------------------------------------------------------- gelu' in eight bits (ME_GG8)

@pytest.mark.parametrize("M_rows,N_out,Kc", [(50432 // 2, 3072, 768), (16384 + 40, 4096, 1024)])
def test_gelu_grad_in_eight_bits(dev, M_rows, N_out, Kc):
    """ME_GG8: the fc1 forward of a bf16 training Block saves gelu'(h) as one byte (-0.13 + q * 1.26 / 255), the fc2 dgrad multiplies by it.
    Saved bytes = the quantised fp32 gelu' of the SAME accumulators the bf16 form rounds to bf16 (so: within one step of the bf16 form's bytes,
    within half a step + the GEMM's own error of the float64 gelu'); the GELU output is bit-identical to the bf16-factor launch; the dgrad
    equals (A W^T) * decode(q) like the bf16-factor launch equals (A W^T) * factor; a ragged last row tile is covered (second case)."""
    if not ops.gemm_takes_gg8(M_rows, N_out, Kc):
        pytest.skip("the resident kernel does not take this shape here")
    a, w, bias = rnd(M_rows, Kc, seed=41).bfloat16(), (0.05 * rnd(N_out, Kc, seed=42)).bfloat16(), 0.1 * rnd(N_out, seed=43)
    ad, wd, bd = a.to(dev), w.to(dev), bias.to(dev)
    q = torch.full((M_rows, N_out), 7, dtype=torch.uint8, device=dev)
    y8 = ops.gemm(ad, wd, bias=bd, act=_capi.ME_ACT_GELU, preact=q, flags=_capi.ME_GEMM_SAVE_GELU_GRAD)
    sav = torch.empty(M_rows, N_out, dtype=torch.bfloat16, device=dev)
    y16 = ops.gemm(ad, wd, bias=bd, act=_capi.ME_ACT_GELU, preact=sav, flags=_capi.ME_GEMM_SAVE_GELU_GRAD)
    assert torch.equal(y8, y16)
    dec = ops.GG8_LO + ops.GG8_STEP * q.float()
    # against the bf16 form (same accumulators): half a step + bf16's own rounding of a value <= 1.13
    assert float((dec - sav.float()).abs().max()) <= 0.5 * ops.GG8_STEP + 2.0 ** -8 + 1e-6
    # against float64 on a row sample
    rows = torch.randperm(M_rows, generator=torch.Generator().manual_seed(5))[:2048]
    h = (a[rows].double() @ w.double().t() + bias.double()).requires_grad_(True)
    bo.gelu_erf(h).sum().backward()
    err = (dec[rows.to(dev)].cpu().double() - h.grad).abs()
    assert float(err.max()) <= 0.5 * ops.GG8_STEP + 6e-3 and float(err.mean()) <= 0.3 * ops.GG8_STEP + 2e-4, (float(err.max()), float(err.mean()))
    # the dgrad half: x decode(q), against the bf16-factor launch fed the decoded factor rounded to bf16 (differs by that rounding only)
    g, wt = rnd(M_rows, Kc, seed=44).bfloat16().to(dev), (0.05 * rnd(N_out, Kc, seed=45)).bfloat16().to(dev)
    d8 = ops.gemm(g, wt, aux=q, flags=_capi.ME_GEMM_AUX_IS_FACTOR)
    lin = g[rows.to(dev)].double().cpu() @ wt.double().cpu().t()
    check_close(d8[rows.to(dev)].float(), lin * dec[rows.to(dev)].double().cpu(), 8e-3, "dgrad x eight-bit factor")
    for _ in range(2):
        assert torch.equal(ops.gemm(g, wt, aux=q, flags=_capi.ME_GEMM_AUX_IS_FACTOR), d8)
    # every code point decodes where the header says (a factor tensor holding all 256 codes)
    allq = (torch.arange(M_rows * N_out, device=dev) % 256).to(torch.uint8).reshape(M_rows, N_out)
    dall = ops.gemm(g, wt, aux=allq, flags=_capi.ME_GEMM_AUX_IS_FACTOR)
    check_close(dall[rows.to(dev)].float(), lin * (ops.GG8_LO + ops.GG8_STEP * allq[rows.to(dev)].double().cpu()), 8e-3, "all 256 codes")


def test_eight_bit_factor_is_refused_off_the_resident_kernel(dev):
    """small problems (another kernel family) and unflagged uses say no instead of mis-reading bytes"""
    a, w = rnd(512, 768, seed=1).bfloat16().to(dev), rnd(3072, 768, seed=2).bfloat16().to(dev)
    q = torch.zeros(512, 3072, dtype=torch.uint8, device=dev)
    assert not ops.gemm_takes_gg8(512, 3072, 768)
    with pytest.raises(_capi.MetaEncError):
        ops.gemm(a, w, aux=q, flags=_capi.ME_GEMM_AUX_IS_FACTOR)
    with pytest.raises(_capi.MetaEncError):
        ops.gemm(a, w, act=_capi.ME_ACT_GELU, preact=q, flags=_capi.ME_GEMM_SAVE_GELU_GRAD)
    a2, w2 = rnd(50432 // 2, 768, seed=1).bfloat16().to(dev), w
    q2 = torch.zeros(50432 // 2, 3072, dtype=torch.uint8, device=dev)
    with pytest.raises(_capi.MetaEncError):
        ops.gemm(a2, w2, aux=q2)                    # eight bits only as the flagged factor
    with pytest.raises(_capi.MetaEncError):
        ops.gemm(a2, w2, preact=q2, act=_capi.ME_ACT_GELU)


def test_block_takes_an_empty_batch(dev):
    """an empty batch (or no tokens) goes through a Block and the encoder like through the reference's: an empty tensor comes back, backward runs"""
    enc = make_encoder(2, 768, 12, dev, torch.bfloat16).train()
    for shape in ((0, 197, 768), (3, 0, 768)):
        x = torch.zeros(*shape, dtype=torch.bfloat16, device=dev, requires_grad=True)
        y = enc(x)
        assert tuple(y.shape) == shape and y.dtype == x.dtype
        y.sum().backward()
        assert x.grad is not None and tuple(x.grad.shape) == shape
    with torch.no_grad():
        assert tuple(enc.eval()(torch.zeros(0, 50, 768, dtype=torch.float32, device=dev)).shape) == (0, 50, 768)
