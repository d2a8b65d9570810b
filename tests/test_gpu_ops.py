"""GPU: per-kernel parity of the HIP path (called through the C ABI wrappers in metatransformer_amd.ops) against the
CPU oracle restatements, on seeded inputs.  fp32 tolerance 1e-3 relative (north star), bf16 stated per test."""

import pytest
import torch

from conftest import TOL_BF16_OP, TOL_F32, check_close, rel_err
from metatransformer_amd import _capi, ops
from oracle import block_oracle as bo
from oracle import tokenizer_oracle as to

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def tol(dt):
    return TOL_F32 if dt == torch.float32 else TOL_BF16_OP


def rnd(*shape, seed=0, scale=1.0):
    return scale * torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


# ----------------------------------------------------------------------------- LayerNorm

@pytest.mark.parametrize("C", [768, 1024, 256, 32, 100])
@pytest.mark.parametrize("dt", DTYPES)
def test_layernorm_fwd(dev, C, dt):
    rows = 77
    x = (rnd(rows, C, seed=1) * 2 + 0.5).to(dt)
    g, b = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    ref = bo.layer_norm(x.float(), g, b, 1e-5)
    y, mean, rstd = ops.layernorm_fwd(x.to(dev), g.to(dev), b.to(dev), 1e-5, dt)
    assert rel_err(y.float(), ref) < (1e-5 if dt == torch.float32 else 8e-3)
    assert rel_err(mean, x.float().mean(-1)) < 1e-5
    assert rel_err(rstd, 1 / torch.sqrt(x.float().var(-1, unbiased=False) + 1e-5)) < 1e-5


@pytest.mark.parametrize("C", [768, 1024, 32, 100])
@pytest.mark.parametrize("dt", DTYPES)
def test_layernorm_bwd(dev, C, dt):
    rows = 1500 if C >= 256 else 50
    x = (rnd(rows, C, seed=1) * 2 + 0.5).to(dt)
    dy = rnd(rows, C, seed=4).to(dt)
    dres = rnd(rows, C, seed=5).to(dt)
    g, b = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    xr = x.float().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    bo.layer_norm(xr, gr, br, 1e-5).backward(dy.float())
    _, mean, rstd = ops.layernorm_fwd(x.to(dev), g.to(dev), b.to(dev), 1e-5, dt)
    dx, dg, db = ops.layernorm_bwd(dy.to(dev), x.to(dev), mean, rstd, g.to(dev), dres.to(dev), dt, True)
    t = 1e-4 if dt == torch.float32 else 1e-2
    assert rel_err(dx.float(), xr.grad + dres.float()) < t
    assert rel_err(dg, gr.grad) < t and rel_err(db, br.grad) < t
    dx2, dg2, _ = ops.layernorm_bwd(dy.to(dev), x.to(dev), mean, rstd, g.to(dev), None, dt, False)
    assert dg2 is None and rel_err(dx2.float(), xr.grad) < t


# ----------------------------------------------------------------------------- GEMM

@pytest.mark.parametrize("M,N,K", [(197 * 2, 768, 768), (256, 2304, 768), (130, 132, 72), (1, 4, 8), (513, 3072, 768),
                                   (300, 768, 3072)])
@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_nt_plain(dev, M, N, K, dt):
    a, b = rnd(M, K, seed=1).to(dt), rnd(N, K, seed=2).to(dt)
    ref = a.double() @ b.double().t()
    c = ops.gemm(a.to(dev), b.to(dev), out_dtype=torch.float32)
    assert rel_err(c, ref) < 1e-5, "fp32-accumulated product of exactly representable inputs"


def test_gemm_layout_is_transpose_detecting(dev):
    """A = I with an asymmetric B catches a row<->col swap of the accumulator layout (CDNA guide G9)."""
    K = 128
    a = torch.eye(K, dtype=torch.bfloat16)
    b = (torch.arange(K * K, dtype=torch.float32).reshape(K, K) % 251 - 125).to(torch.bfloat16)   # b[n,k], asymmetric
    c = ops.gemm(a.to(dev), b.to(dev), out_dtype=torch.float32)          # c[m,n] = sum_k I[m,k] b[n,k] = b[n,m]
    assert torch.equal(c.cpu(), b.float().t().contiguous())


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_nt_epilogues(dev, dt):
    M, N, K = 394, 1536, 256
    a, w = rnd(M, K, seed=1).to(dt), (0.05 * rnd(N, K, seed=2)).to(dt)
    bias, res, cs = 0.1 * rnd(N, seed=3), rnd(M, N, seed=4).to(dt), 1 + 0.2 * rnd(N, seed=5)
    lin = a.double() @ w.double().t() + bias.double()
    t = 1e-5 if dt == torch.float32 else 8e-3          # bf16: output rounding only (inputs exact)
    # bias + GELU with saved pre-activation
    pre = torch.empty(M, N, dtype=dt, device=dev)
    y = ops.gemm(a.to(dev), w.to(dev), bias=bias.to(dev), act=_capi.ME_ACT_GELU, preact=pre)
    assert rel_err(pre.float(), lin) < t and rel_err(y.float(), bo.gelu_erf(lin)) < t
    # bias + colscale + residual, fp32 output
    y = ops.gemm(a.to(dev), w.to(dev), bias=bias.to(dev), colscale=cs.to(dev), residual=res.to(dev), out_dtype=torch.float32)
    assert rel_err(y, lin * cs.double() + res.double()) < 1e-5
    # GELU backward: multiply by gelu'(aux)
    aux = rnd(M, N, seed=6).to(dt)
    xa = aux.double().requires_grad_(True)
    bo.gelu_erf(xa).sum().backward()
    y = ops.gemm(a.to(dev), w.to(dev), aux=aux.to(dev), out_dtype=torch.float32)
    assert rel_err(y, (a.double() @ w.double().t()) * xa.grad) < 1e-5
    # alpha / beta accumulate
    c0 = rnd(M, N, seed=7)
    y = ops.gemm(a.to(dev), w.to(dev), out=c0.clone().to(dev), alpha=0.5, beta=2.0)
    assert rel_err(y, 0.5 * (a.double() @ w.double().t()) + 2.0 * c0.double()) < 1e-5
    # pos-embed broadcast (row modulo) + grouped output rows behind a cls slot
    tps, Bn = 197 - 1, 2
    a2 = rnd(Bn * tps, K, seed=8).to(dt)
    pos = rnd(tps, N, seed=9)
    out = torch.zeros(Bn * 197, N, dtype=torch.float32, device=dev)
    ops.gemm(a2.to(dev), w.to(dev), out=out, residual=pos.to(dev), res_row_mod=tps, out_group=(tps, 197, 1))
    ref = (a2.double() @ w.double().t()).reshape(Bn, tps, N) + pos.double()
    out = out.reshape(Bn, 197, N).cpu()
    assert torch.all(out[:, 0] == 0) and rel_err(out[:, 1:], ref) < 1e-5


@pytest.mark.parametrize("M,N,K", [(394, 1536, 256), (256 * 160, 3072, 256), (256 * 160 + 40, 1032, 128)])
def test_gemm_gelu_grad_pair(dev, M, N, K):
    """ME_GEMM_SAVE_GELU_GRAD / ME_GEMM_AUX_IS_FACTOR: the forward saves gelu'(h) instead of h, the backward multiplies by
    the saved factor (mlp.py:31 and its autograd).  Small shape: generic epilogues; large shapes: the resident kernel's own
    forms (whole and ragged tiles)."""
    dt = torch.bfloat16
    a, w, bias = rnd(M, K, seed=1).to(dt), (0.05 * rnd(N, K, seed=2)).to(dt), 0.1 * rnd(N, seed=3)
    h = (a.double() @ w.double().t() + bias.double()).requires_grad_(True)
    bo.gelu_erf(h).sum().backward()
    sav = torch.empty(M, N, dtype=dt, device=dev)
    y = ops.gemm(a.to(dev), w.to(dev), bias=bias.to(dev), act=_capi.ME_ACT_GELU, preact=sav, flags=_capi.ME_GEMM_SAVE_GELU_GRAD)
    assert rel_err(y.float(), bo.gelu_erf(h.detach())) < 8e-3 and rel_err(sav.float(), h.grad) < 8e-3
    check_close(sav.float().cpu(), h.grad.float(), 8e-3, "saved gelu'")
    # backward half: dA * saved factor, against the product with the factor AS STORED (bf16)
    g, wt = rnd(M, K, seed=4).to(dt), (0.05 * rnd(N, K, seed=5)).to(dt)
    dh = ops.gemm(g.to(dev), wt.to(dev), aux=sav, flags=_capi.ME_GEMM_AUX_IS_FACTOR, out_dtype=dt)
    ref = (g.double() @ wt.double().t()) * sav.double().cpu()
    assert rel_err(dh.float(), ref) < 8e-3
    with pytest.raises(_capi.MetaEncError):
        ops.gemm(a.to(dev), w.to(dev), flags=_capi.ME_GEMM_SAVE_GELU_GRAD)          # needs act = GELU and preact


@pytest.mark.parametrize("T,M,N", [(394, 768, 768), (1000, 3072, 768), (130, 136, 72), (65, 8, 8)])
@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_tn_wgrad(dev, T, M, N, dt):
    a, b = rnd(T, M, seed=1).to(dt), rnd(T, N, seed=2).to(dt)
    ref = a.double().t() @ b.double()
    c = ops.gemm(a.to(dev), b.to(dev), op=_capi.ME_GEMM_TN, out_dtype=torch.float32)
    assert rel_err(c, ref) < 1e-5


@pytest.mark.parametrize("T,M,N", [(256 * 197, 768, 768), (256 * 197, 2304, 768), (8192, 768, 3072), (4096 + 32, 256, 512),
                                   (640, 128, 256), (33 * 197, 768, 768), (4096 + 37, 2304, 768)])      # (the last two: ragged token counts)
@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_tn_fused_bias_gradient(dev, T, M, N, dt):
    """wgrad with the bias gradient (column sums of dY) from the same kernel; falls back to me_colsum when not fusable"""
    dy = rnd(T, M, seed=5).to(dt)
    x = rnd(T, N, seed=6).to(dt)
    dw, db = ops.gemm(dy.to(dev), x.to(dev), op=_capi.ME_GEMM_TN, out_dtype=torch.float32, want_colsum_a=True)
    assert rel_err(dw, dy.double().t() @ x.double()) < 3e-5      # reduction length up to 50 432
    check_close(dw, dy.double().t() @ x.double(), 3e-5, "wgrad")
    ref = dy.double().sum(0)
    assert (db.double().cpu() - ref).abs().max() / ref.abs().max() < 1e-5
    dw2, db2 = ops.gemm(dy.to(dev), x.to(dev), op=_capi.ME_GEMM_TN, out_dtype=torch.float32, want_colsum_a=True)
    assert torch.equal(db, db2) and torch.equal(dw, dw2), "deterministic"


def test_gemm_rejects_bad_args(dev):
    a = torch.zeros(4, 12, device=dev, dtype=torch.bfloat16)
    with pytest.raises(_capi.MetaEncError, match="multiples"):
        ops.gemm(a, a)                       # K=12 not a multiple of 8 bf16
    with pytest.raises(_capi.MetaEncError):
        ops.gemm(a, a.float())               # dtype mismatch
    with pytest.raises(_capi.MetaEncError, match="CUDA"):
        ops.gemm(a.cpu(), a.cpu())


@pytest.mark.parametrize("dt", DTYPES)
def test_colsum_cast_transpose(dev, dt):
    x = rnd(1234, 776, seed=1).to(dt)
    assert rel_err(ops.colsum(x.to(dev)), x.double().sum(0)) < 1e-5
    w = rnd(300, 520, seed=2)
    wt = ops.transpose_cast(w.to(dev), dt)
    assert torch.equal(wt.cpu(), w.t().contiguous().to(dt))
    assert torch.equal(ops.cast(w.to(dev), dt).cpu(), w.to(dt))
    # batched form: four-element lanes where rows / cols / alignment allow it, the element-wise form otherwise (ragged shapes, a
    # source that starts 4 bytes into an allocation), fp32 and bf16 sources
    for src_dt in DTYPES:
        flat = rnd(3 + 768 * 3072, 1, seed=5).to(src_dt).to(dev).view(-1)
        ws = [rnd(768, 3072, seed=6).to(src_dt).to(dev), rnd(300, 520, seed=7).to(src_dt).to(dev), rnd(130, 67, seed=8).to(src_dt).to(dev),
              rnd(64, 4, seed=9).to(src_dt).to(dev), flat[1:1 + 768 * 3072].view(768, 3072), rnd(1000, 768, seed=10).to(src_dt).to(dev)]
        for w2, got in zip(ws, ops.transpose_cast_many(ws, dt)):
            assert torch.equal(got.cpu(), w2.cpu().t().contiguous().to(dt)), tuple(w2.shape)
    pos = rnd(7, 64, seed=3)
    xx = rnd(21, 64, seed=4)
    y = ops.add_rows(xx.to(dev), pos.to(dev))
    assert torch.equal(y.cpu(), xx + pos.repeat(3, 1))


# ----------------------------------------------------------------------------- attention

def attn_ref(qkv, B, N, H, hd, scale):
    q, k, v = qkv.double().reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)) * scale
    p = torch.softmax(s, dim=-1)
    return (p @ v).transpose(1, 2).reshape(B * N, H * hd), torch.logsumexp(s, dim=-1)


ATTN_SHAPES = [(2, 197, 12, 64), (1, 64, 2, 64), (3, 37, 2, 64), (2, 5, 2, 16), (2, 40, 32, 24), (1, 300, 4, 32),
               (1, 130, 2, 128), (1, 1, 1, 64), (1, 256, 2, 64), (2, 224, 3, 32), (2, 33, 1, 8),
               (1, 64, 2, 32), (2, 65, 2, 64), (1, 257, 2, 64), (1, 1568, 2, 64), (2, 512, 3, 64), (1, 400, 2, 64), (1, 513, 1, 64),
               (2, 128, 32, 24), (1, 300, 32, 24), (1, 600, 4, 24),      # Graph: 32 heads x hd 24 on the resident / mid / chunked paths; both sides of the resident-sequence window
               # N <= 64: one wave per (batch, head) (attention_tiny.hip) -- Tabular (16 column tokens), Graph (~50 tokens, 32 x 24), every
               # (N class, head-dim class) pair with ragged N and head dims that are not tile multiples
               (16, 16, 12, 64), (4, 50, 32, 24), (2, 17, 3, 48), (2, 16, 2, 40), (1, 32, 2, 56), (3, 48, 2, 64), (2, 31, 4, 32), (5, 9, 2, 24)]


@pytest.mark.parametrize("B,N,H,hd", ATTN_SHAPES)
@pytest.mark.parametrize("dt", DTYPES)
def test_attention_fwd(dev, B, N, H, hd, dt):
    qkv = rnd(B * N, 3 * H * hd, seed=B * 1000 + N).to(dt)
    scale = hd ** -0.5
    ref, lse_ref = attn_ref(qkv, B, N, H, hd, scale)
    out, lse = ops.attention_fwd(qkv.to(dev), B, N, H, hd, scale, True)
    assert rel_err(out.float(), ref) < (2e-5 if dt == torch.float32 else 1.5e-2)
    assert rel_err(lse, lse_ref) < (1e-5 if dt == torch.float32 else 5e-3)
    # per element: an error confined to the last key chunk / the ragged tail rows of N = 197, 513, 1568 sits in
    # FEW outputs of ordinary size -- the max-abs line above cannot see it under the global scale, the per-element bound can
    check_close(out.float(), ref, 2e-5 if dt == torch.float32 else 1.5e-2, f"attention forward {B}x{N}x{H}x{hd}")
    # ... and row by row: the worst query row of every (batch, head) against that row's own scale
    o4, r4 = out.float().cpu().double().reshape(B, N, H, hd), ref.reshape(B, N, H, hd)
    row_err = (o4 - r4).abs().amax(dim=-1) / r4.abs().amax(dim=-1).clamp_min(1e-3 * float(r4.abs().max()))
    assert float(row_err.max()) < (1e-4 if dt == torch.float32 else 4e-2), (float(row_err.max()), int(row_err.argmax()))


def test_attention_online_softmax_rescale_path(dev):
    """Force the running-max rescale: one key far above the rest, placed in the LAST 64-key tile."""
    B, N, H, hd = 1, 200, 1, 64
    qkv = 0.3 * rnd(B * N, 3 * hd, seed=9)
    qkv[190, hd:2 * hd] = 6.0 * qkv[3, 0:hd] / qkv[3, 0:hd].norm() * 8          # key 190 aligned with query 3
    ref, _ = attn_ref(qkv, B, N, H, hd, hd ** -0.5)
    for dt in DTYPES:
        out, _ = ops.attention_fwd(qkv.to(dt).to(dev), B, N, H, hd, hd ** -0.5, False)
        r2, _ = attn_ref(qkv.to(dt), B, N, H, hd, hd ** -0.5)
        assert rel_err(out.float(), r2) < (2e-5 if dt == torch.float32 else 1.5e-2)


@pytest.mark.parametrize("B,N,H,hd", ATTN_SHAPES)
@pytest.mark.parametrize("dt", DTYPES)
def test_attention_bwd(dev, B, N, H, hd, dt):
    qkv = rnd(B * N, 3 * H * hd, seed=B * 1000 + N + 1).to(dt)
    do = rnd(B * N, H * hd, seed=77).to(dt)
    scale = hd ** -0.5
    qr = qkv.double().requires_grad_(True)
    ref, _ = attn_ref(qr, B, N, H, hd, scale)
    ref.backward(do.double())
    out, lse = ops.attention_fwd(qkv.to(dev), B, N, H, hd, scale, True)
    dqkv = ops.attention_bwd(qkv.to(dev), out, do.to(dev), lse, B, N, H, hd, scale)
    assert rel_err(dqkv.float(), qr.grad) < (5e-5 if dt == torch.float32 else 3e-2)
    # per element, and dQ / dK / dV each against its OWN scale (dV is ~10x larger than dQ at these sizes: a wrong dQ tail would
    # pass a bound taken over the whole [M, 3C] tensor)
    C = H * hd
    whole = float(qr.grad.abs().max())
    for j, nm in enumerate(("dQ", "dK", "dV")):
        got, want = dqkv.float()[:, j * C:(j + 1) * C], qr.grad[:, j * C:(j + 1) * C]
        if float(want.abs().max()) < 1e-3 * whole:       # (N = 1: dQ and dK are identically zero -- no scale of their own)
            assert float((got.cpu().double() - want).abs().max()) <= (5e-5 if dt == torch.float32 else 3e-2) * whole, nm
            continue
        check_close(got, want, 5e-5 if dt == torch.float32 else 3e-2, f"attention backward {nm} {B}x{N}x{H}x{hd}")


@pytest.mark.parametrize("B,N,H,hd", [(96, 197, 12, 64), (70, 100, 12, 48)])
def test_attention_bwd_claimed_items_match_the_static_schedule(dev, B, N, H, hd):
    """the persistent ring backward CLAIMS its items (a returning atomic retired by a hand-counted wait, ~400
    instructions after it was issued).  A launch captured into a hipGraph runs the static schedule: same bits; and the claimed
    form stays bit-identical over many launches next to unrelated traffic on a second stream (workgroups then start unevenly and
    really do end up with different item lists)."""
    dt = torch.bfloat16
    qkv = rnd(B * N, 3 * H * hd, seed=B + N).to(dt).to(dev)
    do = rnd(B * N, H * hd, seed=81).to(dt).to(dev)
    scale = hd ** -0.5
    out, lse = ops.attention_fwd(qkv, B, N, H, hd, scale, True)
    eager = ops.attention_bwd(qkv, out, do, lse, B, N, H, hd, scale).clone()
    assert B * H > 256                                   # more items than CUs: every workgroup draws
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got = ops.attention_bwd(qkv, out, do, lse, B, N, H, hd, scale)      # (first launch on this stream outside the capture)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        got = ops.attention_bwd(qkv, out, do, lse, B, N, H, hd, scale)
    got.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, eager)
    noise = torch.empty(32 << 20, device=dev)
    bad = 0
    for i in range(40):
        if i % 2 == 0:
            with torch.cuda.stream(side):
                noise.add_(1.0)
        bad += int(not torch.equal(ops.attention_bwd(qkv, out, do, lse, B, N, H, hd, scale), eager))
    torch.cuda.synchronize()
    assert bad == 0


@pytest.mark.parametrize("N", [197, 250, 400, 600])
def test_attention_bwd_rows_with_very_negative_lse(dev, N):
    """Query rows whose scaled scores are ALL far below zero (lse < -100): P = exp(s - lse) must still use the exact lse (no
    lower bound on it), and the padded keys of the last key group (zero K rows, s = 0 -> exp(-lse) overflows) must stay harmless.
    Covers the ring (N <= 224), resident, mid and chunked backward kernels."""
    B, H, hd = 2, 2, 64
    dt = torch.bfloat16
    qkv = rnd(B * N, 3 * H * hd, seed=4000 + N)
    q = qkv.view(B, N, 3, H, hd)
    # every key of head 0 gets a common component c; queries 5, 60 and N - 1 point along -c with a large norm: all their scores ~ -180
    c = torch.ones(hd) / hd ** 0.5
    q[:, :, 1, 0, :] = 0.2 * q[:, :, 1, 0, :] + 6.0 * c
    for r in (5, 60, N - 1):
        q[:, r, 0, 0, :] = -240.0 * c + q[:, r, 0, 0, :]
    qkv = qkv.to(dt)
    do = rnd(B * N, H * hd, seed=79).to(dt)
    scale = hd ** -0.5
    qr = qkv.double().requires_grad_(True)
    ref, lse_ref = attn_ref(qr, B, N, H, hd, scale)
    ref.backward(do.double())
    assert float(lse_ref.detach()[:, 0, 5].max()) < -100.0
    out, lse = ops.attention_fwd(qkv.to(dev), B, N, H, hd, scale, True)
    assert rel_err(lse, lse_ref.detach()) < 5e-3
    dqkv = ops.attention_bwd(qkv.to(dev), out, do.to(dev), lse, B, N, H, hd, scale)
    assert torch.isfinite(dqkv.float()).all()
    assert rel_err(dqkv.float(), qr.grad) < 3e-2
    # the rows in question on their own (dq of a row with a wrong P is off by O(1) of ITS scale, invisible in the global maximum)
    got = dqkv.float().cpu().view(B, N, 3, H, hd)
    want = qr.grad.float().view(B, N, 3, H, hd)
    # (bound: the keys share a large common component here, so the bf16 rounding of dS -- whose row sums cancel exactly only in
    # exact arithmetic -- shows up as a common offset of a few % in dq; a P taken against a wrong lse is off by orders of magnitude)
    for r in (5, 60, N - 1):
        assert rel_err(got[:, r, 0, 0], want[:, r, 0, 0]) < 0.15, r


# the persistent ring / streaming kernels: more (batch, head[, query block]) items than CUs (a workgroup walks several items: ring
# swap, prefetch chain, loader stream across item boundaries), head_dim below the template width (zero chunks through the DMA
# descriptor), every sub-tile count of the ring form, 13 / 14 compute waves, sequence lengths around the chunk size of the stream
RING_SHAPES = [(40, 197, 12, 64), (30, 100, 12, 48), (2, 209, 3, 64), (3, 224, 2, 64), (2, 96, 2, 64), (2, 129, 2, 40),
               (2, 161, 3, 64), (40, 300, 12, 64), (3, 1000, 2, 48), (2, 225, 2, 64), (2, 384, 2, 64), (2, 385, 3, 64),
               (9, 640, 4, 64), (300, 66, 2, 32),
               # streaming backward (N >= 560, 32 < hd <= 64): several items per workgroup, the shortest sequence it takes, four
               # loader waves (176-row blocks), a ragged last block and chunk, head_dim below the template width
               (12, 592, 12, 64), (2, 560, 2, 64), (2, 673, 3, 64), (1, 3136, 2, 64), (3, 577, 2, 40),
               # the 32-key dK / dV kernel (key blocks that fill seven waves: 192 < keys per block <= 224): full last block, a last block
               # whose second key tiles lie past N, a last block of ONE key, more items than CUs with a short head_dim
               (2, 896, 2, 64), (1, 870, 3, 64), (1, 1345, 1, 64), (24, 600, 12, 48)]


@pytest.mark.parametrize("B,N,H,hd", RING_SHAPES)
def test_attention_persistent_kernels_fwd_bwd(dev, B, N, H, hd):
    dt = torch.bfloat16
    qkv = rnd(B * N, 3 * H * hd, seed=B * 100 + N).to(dt)
    do = rnd(B * N, H * hd, seed=78).to(dt)
    scale = hd ** -0.5
    qr = qkv.double().requires_grad_(True)
    ref, lse_ref = attn_ref(qr, B, N, H, hd, scale)
    ref.backward(do.double())
    out, lse = ops.attention_fwd(qkv.to(dev), B, N, H, hd, scale, True)
    assert rel_err(out.float(), ref.detach()) < 1.5e-2
    assert rel_err(lse, lse_ref.detach()) < 5e-3
    # every item on its own (an item that took another item's K / V, or a stale ring half, is far off while the global error stays small)
    per_item = (out.float().cpu().view(B, N, H * hd) - ref.detach().float().view(B, N, H * hd)).abs().amax(dim=(1, 2))
    assert float(per_item.max()) < 0.05 * float(ref.detach().abs().max()), int(per_item.argmax())
    dqkv = ops.attention_bwd(qkv.to(dev), out, do.to(dev), lse, B, N, H, hd, scale)
    assert rel_err(dqkv.float(), qr.grad) < 3e-2
    per_item = (dqkv.float().cpu().view(B, N, -1) - qr.grad.float().view(B, N, -1)).abs().amax(dim=(1, 2))
    assert float(per_item.max()) < 0.08 * float(qr.grad.abs().max()), int(per_item.argmax())
    # run to run identical (no atomics, fixed reduction order)
    out2, _ = ops.attention_fwd(qkv.to(dev), B, N, H, hd, scale, True)
    assert torch.equal(out, out2)
    assert torch.equal(dqkv, ops.attention_bwd(qkv.to(dev), out, do.to(dev), lse, B, N, H, hd, scale))


@pytest.mark.parametrize("B,N,H,hd", [(32, 1568, 16, 64), (64, 592, 12, 64), (128, 512, 16, 64)])
def test_attention_full_batch_of_configs_3_4_5(dev, B, N, H, hd):
    """the attention launches of BASELINE configs 5 / 4 / 3 at the batch bench.py runs them at (the CPU-oracle tests above cut the batch):
    forward, lse and every gradient against float64 attention computed on the GPU one batch item at a time, per item -- the streaming
    kernels then walk 14 / 9 items per workgroup (config 5: 3 584 key blocks on 256 CUs, the 32-key dK / dV kernel)"""
    dt = torch.bfloat16
    C = H * hd
    g = torch.Generator().manual_seed(B + N)
    qkv = torch.randn(B * N, 3 * C, generator=g).to(dt).to(dev)
    do = torch.randn(B * N, C, generator=g).to(dt).to(dev)
    scale = hd ** -0.5
    out, lse = ops.attention_fwd(qkv, B, N, H, hd, scale, True)
    dqkv = ops.attention_bwd(qkv, out, do, lse, B, N, H, hd, scale)
    worst_o = worst_g = worst_l = 0.0
    for b in range(B):
        x = qkv[b * N:(b + 1) * N].double().requires_grad_(True)
        q, k, v = (x[:, i * C:(i + 1) * C].reshape(N, H, hd).transpose(0, 1) for i in range(3))
        sc = (q @ k.transpose(1, 2)) * scale
        ref_lse = torch.logsumexp(sc, dim=-1)                       # [H, N]
        ref = (torch.softmax(sc, dim=-1) @ v).transpose(0, 1).reshape(N, C)
        ref.backward(do[b * N:(b + 1) * N].double())
        so, sg = float(ref.detach().abs().max()), float(x.grad.abs().max())
        worst_o = max(worst_o, float((out[b * N:(b + 1) * N].double() - ref.detach()).abs().max()) / so)
        worst_g = max(worst_g, float((dqkv[b * N:(b + 1) * N].double() - x.grad).abs().max()) / sg)
        worst_l = max(worst_l, float((lse[b].double() - ref_lse.detach()).abs().max()))
    # per item max-abs error relative to the item's own scale: bf16 operands and probabilities (2^-9), fp32 accumulation
    assert worst_o < 2e-2 and worst_g < 3e-2 and worst_l < 2e-2, (worst_o, worst_g, worst_l)
    assert torch.equal(dqkv, ops.attention_bwd(qkv, out, do, lse, B, N, H, hd, scale))


@pytest.mark.parametrize("B,N,H,hd", [(4, 130, 8, 32), (2, 40, 32, 24), (1, 300, 2, 64)])
@pytest.mark.parametrize("dt", DTYPES)
def test_attention_dropout(dev, B, N, H, hd, dt):
    """training-mode attn_drop: masks are a hash of (seed, b, h, q, k) -- reproducible, unbiased, and the backward uses
    the SAME masks (checked as a directional derivative of the seeded forward, fp32)"""
    p, seed = 0.25, 1234567
    scale = hd ** -0.5
    qkv = (0.5 * rnd(B * N, 3 * H * hd, seed=21)).to(dt).to(dev)
    o0, _ = ops.attention_fwd(qkv, B, N, H, hd, scale, False)
    o1, lse = ops.attention_fwd(qkv, B, N, H, hd, scale, True, p_drop=p, seed=seed)
    o2, _ = ops.attention_fwd(qkv, B, N, H, hd, scale, False, p_drop=p, seed=seed)
    o3, _ = ops.attention_fwd(qkv, B, N, H, hd, scale, False, p_drop=p, seed=seed + 1)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    # unbiased: E[dropout(P)] = P, so the mean over many (query, head) rows of (o_drop - o) is ~0 relative to its spread
    d = (o1.float() - o0.float())
    assert d.abs().max() > 0 and abs(d.mean().item()) < 6 * d.std().item() / (d.numel() ** 0.5) + 1e-4
    if dt == torch.float32:
        g = torch.Generator().manual_seed(5)
        do = torch.randn(B * N, H * hd, generator=g).to(dev)
        dirn = torch.randn(B * N, 3 * H * hd, generator=g).to(dev)
        dqkv = ops.attention_bwd(qkv, o1, do, lse, B, N, H, hd, scale, p_drop=p, seed=seed)
        eps = 1e-2
        fp, _ = ops.attention_fwd(qkv + eps * dirn, B, N, H, hd, scale, False, p_drop=p, seed=seed)
        fm, _ = ops.attention_fwd(qkv - eps * dirn, B, N, H, hd, scale, False, p_drop=p, seed=seed)
        num = ((fp.double() - fm.double()) * do.double()).sum() / (2 * eps)
        ana = (dqkv.double() * dirn.double()).sum()
        assert abs(num - ana) < 2e-3 * max(1.0, abs(ana)), (num.item(), ana.item())


# ----------------------------------------------------------------------------- tokenizer kernels

@pytest.mark.parametrize("geom", [((2, 3, 64, 48), (1, 16, 16, 1, 16, 16)), ((2, 1, 40, 57), (1, 16, 16, 1, 10, 10)),
                                  ((1, 3, 4, 32, 32), (2, 16, 16, 2, 16, 16)),
                                  ((3, 2, 30, 27), (1, 6, 6, 1, 5, 7)),            # kw % 4 != 0: the element-wise kernel
                                  ((2, 3, 36, 44), (1, 8, 12, 1, 7, 4))])          # quads, aligned sources, overlapping rows
def test_patchify_bit_exact(dev, geom):
    shape, (kt, kh, kw, st, sh, sw) = geom
    x = rnd(*shape, seed=3)
    cols, tps = ops.patchify(x.to(dev), kt, kh, kw, st, sh, sw, torch.float32)
    ref = to.patchify_2d(x, kh, kw, sh, sw) if len(shape) == 4 else to.patchify_3d(x, kt, kh, kw)
    assert tps == ref.shape[1]
    assert torch.equal(cols.cpu().reshape(ref.shape), ref), "patch gather must be bit-exact"
    # every dtype pair the tokenizers use: bf16 pixels -> bf16, fp32 pixels -> bf16 (rounded once, as a cast would)
    xb = x.bfloat16()
    cb, _ = ops.patchify(xb.to(dev), kt, kh, kw, st, sh, sw, torch.bfloat16)
    refb = to.patchify_2d(xb.float(), kh, kw, sh, sw) if len(shape) == 4 else to.patchify_3d(xb.float(), kt, kh, kw)
    assert torch.equal(cb.cpu().float().reshape(refb.shape), refb)
    cfb, _ = ops.patchify(x.to(dev), kt, kh, kw, st, sh, sw, torch.bfloat16)
    assert torch.equal(cfb.cpu().float().reshape(refb.shape), refb)
    # scatter-add back: adjoint of the gather
    d = rnd(*cols.shape, seed=4)
    dx = ops.unpatchify_add(d.to(dev), shape, kt, kh, kw, st, sh, sw)
    xr = x.clone().requires_grad_(True)
    r = to.patchify_2d(xr, kh, kw, sh, sw) if len(shape) == 4 else to.patchify_3d(xr, kt, kh, kw)
    r.backward(d.reshape(r.shape))
    assert rel_err(dx, xr.grad) < 1e-6


def test_adamw_matches_torch(dev):
    p = rnd(5000, seed=1); g = rnd(5000, seed=2)
    ref = torch.nn.Parameter(p.clone()); ref.grad = g.clone() * 0.5
    opt = torch.optim.AdamW([ref], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    pd, m, v = p.clone().to(dev), torch.zeros(5000, device=dev), torch.zeros(5000, device=dev)
    for step in (1, 2, 3):
        opt.step()
        ops.adamw_step(pd, g.to(dev), m, v, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05, step=step, grad_scale=0.5)
    assert rel_err(pd, ref.data) < 1e-5


# ---------------------------------------------------------------- fp8 (e4m3) attention forward, BASELINE config 5
def _attn_ref(qkv, B, N, H, hd, scale):
    q, k, v = qkv.double().reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)) * scale
    lse = torch.logsumexp(s, dim=-1)
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B * N, H * hd)
    return o, lse


@pytest.mark.parametrize("B,N,H", [(2, 1568, 4), (1, 197, 3), (3, 64, 2), (1, 130, 1), (2, 700, 2)])
def test_attention_fwd_fp8_vs_fp64(dev, B, N, H):
    """e4m3 Q / K / V / P on the block-scaled MFMA against an fp64 evaluation of modeling_finetune.py:172-195.  The
    reference has no fp8 path ("parity unpinned" for this row); the bound is what e4m3's 3 mantissa bits allow: every
    product p*v carries two roundings of 3.6 % rms each, and on i.i.d. data the output is a random-walk sum of such terms,
    so the error is ~5 % of the typical output whatever N is.  Measured on MI355X: relative rms error 4.6e-2 .. 5.2e-2,
    max error 5e-2 .. 9.3e-2 of max|out| -> bounds 7e-2 rms, 1.5e-1 max (the bf16 kernels: 1.5e-2 max)."""
    hd = 64
    g = torch.Generator().manual_seed(B * 1000 + N)
    qkv = torch.randn(B * N, 3 * H * hd, generator=g).bfloat16()
    scale = hd ** -0.5
    o_ref, lse_ref = _attn_ref(qkv.float(), B, N, H, hd, scale)
    o, lse = ops.attention_fwd(qkv.to(dev), B, N, H, hd, scale, need_lse=True, fp8=True)
    assert o.dtype == torch.bfloat16 and lse.shape == (B, H, N)
    rms = float((o.double().cpu() - o_ref).norm() / o_ref.norm())
    print(f"fp8 attention B={B} N={N} H={H}: rel rms {rms:.2e}, max-norm {rel_err(o.float(), o_ref):.2e}")
    assert rms < 7e-2 and rel_err(o.float(), o_ref) < 1.5e-1
    assert (lse.double().cpu() - lse_ref).abs().max() < 5e-2          # log-sum-exp of scores quantised to 3 mantissa bits
    # the bf16 kernel on the same input, for scale
    o16, _ = ops.attention_fwd(qkv.to(dev), B, N, H, hd, scale, need_lse=False)
    assert rel_err(o16.float(), o_ref) < 1.5e-2


def _attn_fp8_emulated(qkv, B, N, H, scale):
    """The fp8 kernel's arithmetic restated in torch (attention_fp8.hip): per-tensor absmax -> 240 scales, Q / K / V rounded to
    OCP e4m3 (torch.float8_e4m3fn: the same round-to-nearest-even), exact score products, online softmax over 64-key blocks
    in the log2 domain with the running maximum of THAT block, probabilities stored as e4m3(2^7 p) while the row sum keeps the
    unrounded values, fp64 accumulation everywhere else."""
    hd, C = 64, H * 64
    x = qkv.float().reshape(B, N, 3, H, hd)
    q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)      # [B, H, N, hd]
    aq, ak, av = (float(t.abs().max()) for t in (q, k, v))
    f8 = lambda t: t.to(torch.float8_e4m3fn).double()          # noqa: E731
    q8, k8, v8 = f8(q * (240.0 / aq)), f8(k * (240.0 / ak)), f8(v * (240.0 / av))
    c2 = scale * (aq / 240.0) * (ak / 240.0) * 1.4426950408889634
    m = torch.full((B, H, N), -1e30, dtype=torch.float64)
    lsum = torch.zeros(B, H, N, dtype=torch.float64)
    o = torch.zeros(B, H, N, hd, dtype=torch.float64)
    for k0 in range(0, N, 64):
        sblk = (q8 @ k8[:, :, k0:k0 + 64].transpose(-2, -1)) * c2                      # [B, H, N, <=64], log2 units
        mn = torch.maximum(m, sblk.max(dim=-1).values)
        alpha = torch.exp2(m - mn)
        pblk = torch.exp2(sblk - mn.unsqueeze(-1) + 7.0)
        lsum = lsum * alpha + pblk.sum(-1)
        o = o * alpha.unsqueeze(-1) + f8(pblk.float()) @ v8[:, :, k0:k0 + 64]
        m = mn
    out = o * ((av / 240.0) / lsum).unsqueeze(-1)
    lse = (m - 7.0 + torch.log2(lsum)) * 0.6931471805599453
    return out.permute(0, 2, 1, 3).reshape(B * N, C), lse


@pytest.mark.parametrize("B,N,H", [(2, 1568, 2), (1, 197, 3), (1, 130, 1), (2, 700, 2)])
def test_attention_fwd_fp8_vs_emulated_e4m3(dev, B, N, H):
    """the 7 % / 15 % bounds against fp64 are e4m3's own error and would not notice a mis-scaled P or a
    wrong key permutation that costs a few per cent more.  Against the kernel's OWN arithmetic restated in torch (same
    roundings to e4m3, same block-wise running maximum) what is left is fp32 accumulation and v_exp_f32 against exp2 plus the odd
    probability that rounds the other way at an e4m3 tie: bound 1e-2 of max|out| (measured ~1e-3)."""
    hd = 64
    g = torch.Generator().manual_seed(B * 77 + N)
    qkv = torch.randn(B * N, 3 * H * hd, generator=g).bfloat16()
    scale = hd ** -0.5
    o_emu, lse_emu = _attn_fp8_emulated(qkv, B, N, H, scale)
    o, lse = ops.attention_fwd(qkv.to(dev), B, N, H, hd, scale, need_lse=True, fp8=True)
    err = rel_err(o.float(), o_emu)
    rms = float((o.double().cpu() - o_emu).norm() / o_emu.norm())
    print(f"fp8 attention vs emulated e4m3, B={B} N={N} H={H}: max-norm {err:.2e}, rel rms {rms:.2e}")
    assert err < 1e-2 and rms < 5e-3
    assert (lse.double().cpu() - lse_emu).abs().max() < 1e-3


def test_attention_fwd_fp8_peaked_rows_and_rescale(dev):
    """a key that dominates its row late in the sequence forces the running-max rescale (cdna_hip_programming.md rule 26)"""
    B, N, H, hd = 1, 512, 1, 64
    g = torch.Generator().manual_seed(3)
    qkv = (0.3 * torch.randn(B * N, 3 * hd, generator=g))
    qkv[:, :hd][100] = 4.0                      # query 100 ...
    qkv[:, hd:2 * hd][400] = 4.0                # ... meets its spike at key 400 (block 6): score 16 * 8 * scale
    qkv = qkv.bfloat16()
    o_ref, _ = _attn_ref(qkv.float(), B, N, H, hd, hd ** -0.5)
    o, _ = ops.attention_fwd(qkv.to(dev), B, N, H, hd, hd ** -0.5, need_lse=False, fp8=True)
    assert rel_err(o.float(), o_ref) < 6e-2
    assert (o[100].float().cpu() - o_ref[100].float()).abs().max() < 6e-2 * o_ref.abs().max()


def test_attention_fp8_rejects_what_it_does_not_implement(dev):
    from metatransformer_amd import MetaEncError
    qkv = torch.randn(2 * 64, 3 * 2 * 32, device=dev).bfloat16()
    with pytest.raises(MetaEncError):
        ops.attention_fwd(qkv, 2, 64, 2, 32, 0.1, need_lse=False, fp8=True)          # head_dim 32
    with pytest.raises(MetaEncError):
        ops.attention_fwd(qkv.float(), 2, 64, 1, 64, 0.1, need_lse=False, fp8=True)   # fp32 qkv


@pytest.mark.parametrize("M,N,K", [(256 * 90 + 77, 768, 256), (256 * 88 + 200, 768, 768), (256 * 30 + 130, 3072, 256)])
def test_gemm_resident_half_items_every_epilogue(dev, M, N, K):
    """Tile quantisation on the resident NT kernel: when the last round holds at most half the CUs' worth of tiles, those tiles
    run as two 128-row items each (gemm3.hip, g3_phase<.., HALF>).  Shapes chosen so that this happens (273 / 267 / 372 tiles on
    256 CUs) with a ragged last tile row -- 77 rows: its second half is EMPTY; 200 rows: its second half is partial.  The bias-only
    and folded-LayerNorm forms carry the items (launch3r: HI); the other epilogues run the same shapes as whole tiles -- every form,
    every output element checked."""
    dt = torch.bfloat16
    a, w, bias = rnd(M, K, seed=1).to(dt), (0.05 * rnd(N, K, seed=2)).to(dt), 0.1 * rnd(N, seed=3)
    lin = a.double() @ w.double().t() + bias.double()
    ad, wd, bd = a.to(dev), w.to(dev), bias.to(dev)
    check_close(ops.gemm(ad, wd, bias=bd).float(), lin, 8e-3, "bias")
    check_close(ops.gemm(ad, wd, bias=bd, act=_capi.ME_ACT_GELU).float(), bo.gelu_erf(lin), 8e-3, "gelu")
    res = rnd(M, N, seed=4).to(dt)
    check_close(ops.gemm(ad, wd, bias=bd, residual=res.to(dev)).float(), lin + res.double(), 8e-3, "residual")
    pre = torch.full((M, N), float("nan"), dtype=dt, device=dev)
    y = ops.gemm(ad, wd, bias=bd, act=_capi.ME_ACT_GELU, preact=pre)
    check_close(pre.float(), lin, 8e-3, "saved pre-activation")
    check_close(y.float(), bo.gelu_erf(lin), 8e-3, "gelu next to the saved pre-activation")
    h = lin.clone().requires_grad_(True)
    bo.gelu_erf(h).sum().backward()
    sav = torch.full((M, N), float("nan"), dtype=dt, device=dev)
    y = ops.gemm(ad, wd, bias=bd, act=_capi.ME_ACT_GELU, preact=sav, flags=_capi.ME_GEMM_SAVE_GELU_GRAD)
    check_close(sav.float(), h.grad, 8e-3, "saved gelu'")
    check_close(y.float(), bo.gelu_erf(lin), 8e-3, "gelu next to the saved gelu'")
    fac = rnd(M, N, seed=6).to(dt)
    check_close(ops.gemm(ad, wd, aux=fac.to(dev), flags=_capi.ME_GEMM_AUX_IS_FACTOR).float(), (a.double() @ w.double().t()) * fac.double(), 8e-3, "* factor")
    st = ops.row_stats(ad, 1e-5)
    want = (st[:, 0:1].double().cpu() * (a.double() @ w.double().t()) + st[:, 1:2].double().cpu() * w.double().sum(1)[None, :] + bias.double())
    check_close(ops.gemm(ad, wd, bias=bd, row_affine=st, col_shift=w.float().sum(1).to(dev)).float(), want, 8e-3, "folded LayerNorm form")
    # run to run: the items are CLAIMED (which CU computes which is not fixed), every element's reduction order is -- same bits
    y1 = ops.gemm(ad, wd, bias=bd, residual=res.to(dev))
    for _ in range(3):
        assert torch.equal(ops.gemm(ad, wd, bias=bd, residual=res.to(dev)), y1)
    # claimed against STATIC schedule (the ticket of the claimed schedule travels through a hand-counted wait): a launch
    # captured into a hipGraph always runs the static one -- same bits, every form that has a resident kernel
    resd, facd, csd, outs = res.to(dev), fac.to(dev), w.float().sum(1).to(dev), {}
    forms = {"bias": lambda o: ops.gemm(ad, wd, bias=bd, out=o), "gelu": lambda o: ops.gemm(ad, wd, bias=bd, act=_capi.ME_ACT_GELU, out=o),
             "residual": lambda o: ops.gemm(ad, wd, bias=bd, residual=resd, out=o),
             "factor": lambda o: ops.gemm(ad, wd, aux=facd, flags=_capi.ME_GEMM_AUX_IS_FACTOR, out=o),
             "folded": lambda o: ops.gemm(ad, wd, bias=bd, row_affine=st, col_shift=csd, out=o)}
    eager = {k: f(torch.empty(M, N, dtype=dt, device=dev)).clone() for k, f in forms.items()}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for k, f in forms.items():
            outs[k] = torch.empty(M, N, dtype=dt, device=dev)
            f(outs[k])                                 # (first launch on this stream outside the capture)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for k, f in forms.items():
            f(outs[k])
    for o in outs.values():
        o.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    for k in forms:
        assert torch.equal(outs[k], eager[k]), k
