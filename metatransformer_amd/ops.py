"""Thin Python wrappers over the C ABI: torch tensors in, device pointers out.

torch is plumbing here (device memory, streams); every computation happens in libmetaenc.so.
All wrappers require CUDA(ROCm) tensors and raise if handed CPU tensors -- no fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _capi
from ._capi import GemmDesc, MetaEncError, check, dtype_code, ptr, stream_ptr

# bumped by every libmetaenc op that mutates parameters in place behind autograd's back (the fused optimizer):
# the per-block weight caches key on it, because raw-pointer writes do not bump tensor._version.
WEIGHT_EPOCH = 0

# bench.py sets this to a list to bracket every me_gemm launch with events on the launch stream:
# entries are (op, ab_dtype_code, M, N, K, start_event, end_event).
def gemm_profile(on: bool) -> None:
    """Start (and reset) / stop the library's per-launch HIP-event timing of me_gemm (also inside me_block_fwd/bwd)."""
    check(_capi.load().me_gemm_profile_enable(1 if on else 0), "me_gemm_profile_enable")


def gemm_profile_read(max_records: int = 1 << 16, with_plan: bool = False):
    """-> [(op, ab_dtype, M, N, K, ms[, plan])] in call order since gemm_profile(True); waits for the recorded events.
    plan (with_plan=True): me_gemm_profile_rec.plan -- kernel family / split-K parts of a GEMM record."""
    lib = _capi.load()
    buf = (_capi.GemmProfileRec * max_records)()
    n = lib.me_gemm_profile_read(buf, max_records)
    if with_plan:
        return [(r.op, r.ab_dtype, r.M, r.N, r.K, r.ms, r.plan) for r in buf[:min(n, max_records)]]
    return [(r.op, r.ab_dtype, r.M, r.N, r.K, r.ms) for r in buf[:min(n, max_records)]]


def _req(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise MetaEncError(f"{name}: expected a CUDA/ROCm tensor (libmetaenc has no CPU path)")
    if not t.is_contiguous():
        raise MetaEncError(f"{name}: tensor must be contiguous")
    return t


def _f32(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """Small per-channel vectors (LayerNorm affine, biases) are consumed as fp32."""
    if t is None:
        return None
    return t if t.dtype == torch.float32 else t.float()


# ----------------------------------------------------------------------------- LayerNorm

def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                  out_dtype: torch.dtype, save_stats: bool = True):
    lib = _capi.load()
    _req(x, "x")
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    g, b = _f32(gamma).contiguous(), _f32(beta).contiguous()
    check(lib.me_layernorm_fwd(ptr(x), dtype_code(x.dtype), ptr(g), ptr(b), ptr(y), dtype_code(out_dtype),
                               ptr(mean), ptr(rstd), rows, C, float(eps), stream_ptr()), "me_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, gamma: torch.Tensor,
                  dres: Optional[torch.Tensor], dx_dtype: torch.dtype, need_affine: bool,
                  affine_accum: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """affine_accum = (dgamma, dbeta) fp32 [C] buffers to ACCUMULATE the affine gradients into (then returned as None)."""
    lib = _capi.load()
    _req(dy, "dy"); _req(x, "x")
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty(x.shape, dtype=dx_dtype, device=x.device)
    g = _f32(gamma).contiguous()
    accumulate = 0
    if need_affine and affine_accum is not None:
        dgamma, dbeta = affine_accum
        accumulate = 1
    else:
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device) if need_affine else None
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device) if need_affine else None
    ws = None
    if need_affine:
        ws = torch.empty(lib.me_layernorm_bwd_workspace(C), dtype=torch.uint8, device=x.device)
    check(lib.me_layernorm_bwd(ptr(dy), dtype_code(dy.dtype), ptr(x), dtype_code(x.dtype), ptr(mean), ptr(rstd), ptr(g),
                               ptr(dres), dtype_code(dres.dtype) if dres is not None else 0,
                               ptr(dx), dtype_code(dx_dtype), ptr(dgamma), ptr(dbeta), accumulate, rows, C, ptr(ws),
                               stream_ptr()), "me_layernorm_bwd")
    if accumulate:
        return dx, None, None
    return dx, dgamma, dbeta


# ----------------------------------------------------------------------------- GEMM

def gemm(a: torch.Tensor, b: torch.Tensor, *, op: int = _capi.ME_GEMM_NT, out: Optional[torch.Tensor] = None,
         out_dtype: Optional[torch.dtype] = None, bias: Optional[torch.Tensor] = None, act: int = _capi.ME_ACT_NONE,
         residual: Optional[torch.Tensor] = None, res_row_mod: int = 0, preact: Optional[torch.Tensor] = None,
         aux: Optional[torch.Tensor] = None, colscale: Optional[torch.Tensor] = None, alpha: float = 1.0,
         beta: float = 0.0, out_rows: Optional[int] = None, out_group: Tuple[int, int, int] = (0, 0, 0),
         want_colsum_a: bool = False, colsum_out: Optional[torch.Tensor] = None, flags: int = 0,
         row_affine: Optional[torch.Tensor] = None, col_shift: Optional[torch.Tensor] = None, want_row_stats: bool = False,
         row_parts: Optional[torch.Tensor] = None, row_eps: float = 0.0):
    """NT: out[M,N] = a[M,K] @ b[N,K]^T ;  TN: out[M,N] = a[K,M]^T @ b[K,N]; fused epilogue per include/metaenc.h.
    want_colsum_a (TN): also return sum_k a[k, :] (fp32 [M]) -- the bias gradient that goes with a weight gradient --
    from the same kernel when the library can fuse it, else from me_colsum; the result is then (out, colsum).
    colsum_out: fp32 [M] buffer for it, accumulated with the same beta as out (only used when the kernel fuses it).
    flags: ME_GEMM_SAVE_GELU_GRAD (preact receives gelu'(pre-activation)) / ME_GEMM_AUX_IS_FACTOR (multiply by aux itself).
    want_row_stats (NT + residual): also return the per-row partial statistics of the OUTPUT, [N / 256, M, 2] fp32 (mean, M2) over
    256-column groups (me_gemm_desc.row_stats; fold with row_stats_combine) -- None when the kernel for this problem cannot emit
    them (me_gemm_emits_row_stats); the result is then (out, partials or None).
    row_parts (NT, with col_shift, instead of row_affine): such partials of THIS launch's A operand, [K / 256, M, 2] -- the kernel forms the
    folded LayerNorm(K, row_eps) pairs itself (me_gemm_desc.row_parts); raises where me_gemm_takes_row_parts says no (gemm_takes_row_parts)."""
    lib = _capi.load()
    _req(a, "a"); _req(b, "b")
    if a.dtype != b.dtype:
        raise MetaEncError(f"gemm: operand dtypes differ ({a.dtype} vs {b.dtype})")
    a2 = a.reshape(-1, a.shape[-1])
    b2 = b.reshape(-1, b.shape[-1])
    if op == _capi.ME_GEMM_NT:
        M, K = a2.shape
        N, Kb = b2.shape
    else:
        K, M = a2.shape
        Kb, N = b2.shape
    if K != Kb:
        raise MetaEncError(f"gemm: reduction dims differ ({K} vs {Kb})")
    if out is None:
        out = torch.empty((out_rows if out_rows is not None else M, N), dtype=out_dtype or a.dtype, device=a.device)
    _req(out, "out")
    d = GemmDesc()
    d.op, d.ab_dtype = op, dtype_code(a.dtype)
    d.M, d.N, d.K = M, N, K
    d.A, d.lda = ptr(a2), a2.stride(0)
    d.B, d.ldb = ptr(b2), b2.stride(0)
    d.C, d.ldc, d.c_dtype = ptr(out), out.stride(-2), dtype_code(out.dtype)
    d.act, d.alpha, d.beta = act, alpha, beta
    d.flags = flags
    keep = []
    if row_affine is not None:        # folded LayerNorm: [M, 2] fp32 pairs from row_stats + s [N] (include/metaenc.h)
        if row_affine.dtype != torch.float32 or row_affine.shape != (M, 2) or col_shift is None or col_shift.numel() != N:
            raise MetaEncError("gemm: row_affine must be [M, 2] float32 (ops.row_stats) and col_shift [N]")
        col_shift = _f32(col_shift).contiguous(); keep.append(col_shift)
        d.row_affine, d.col_shift = ptr(_req(row_affine, "row_affine")), ptr(col_shift)
    if row_parts is not None:
        if row_affine is not None or col_shift is None or col_shift.numel() != N or row_parts.dtype != torch.float32 or \
                K % 256 or tuple(row_parts.shape) != (K // 256, M, 2):
            raise MetaEncError("gemm: row_parts must be [K / 256, M, 2] float32 partials (gemm(..., want_row_stats=True)), with col_shift [N], without row_affine")
        col_shift = _f32(col_shift).contiguous(); keep.append(col_shift)
        d.row_parts, d.row_nparts, d.row_eps, d.col_shift = ptr(_req(row_parts, "row_parts")), K // 256, float(row_eps), ptr(col_shift)
    if bias is not None:
        bias = _f32(bias).contiguous(); keep.append(bias)
        d.bias = ptr(bias)
    if colscale is not None:
        colscale = _f32(colscale).contiguous(); keep.append(colscale)
        d.colscale = ptr(colscale)
    if preact is not None:
        _req(preact, "preact")
        # (uint8 = ME_GG8: gelu' in eight bits, -0.13 + q * 1.26 / 255 -- with ME_GEMM_SAVE_GELU_GRAD, where gemm_takes_gg8 says yes)
        d.preact, d.ldpre, d.preact_dtype = ptr(preact), preact.stride(-2), _capi.ME_GG8 if preact.dtype == torch.uint8 else dtype_code(preact.dtype)
    if aux is not None:
        _req(aux, "aux")
        d.aux, d.ldaux, d.aux_dtype = ptr(aux), aux.stride(-2), _capi.ME_GG8 if aux.dtype == torch.uint8 else dtype_code(aux.dtype)
    if residual is not None:
        _req(residual, "residual")
        d.residual, d.ldres, d.res_dtype = ptr(residual), residual.stride(-2), dtype_code(residual.dtype)
        d.res_row_mod = res_row_mod
    d.out_group_rows, d.out_group_stride, d.out_row_offset = out_group
    cs = None
    if want_colsum_a:
        if op != _capi.ME_GEMM_TN:
            raise MetaEncError("gemm: want_colsum_a is defined for ME_GEMM_TN")
        if lib.me_gemm_fuses_colsum(ctypes.byref(d)):
            cs = colsum_out if colsum_out is not None else torch.empty(M, dtype=torch.float32, device=a.device)
            keep.append(cs)
            d.colsum_a = ptr(cs)

    ws_bytes = lib.me_gemm_workspace_bytes(ctypes.byref(d))
    if ws_bytes:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device); keep.append(ws)
        d.workspace, d.workspace_bytes = ptr(ws), ws_bytes
    partials = None
    if want_row_stats and lib.me_gemm_emits_row_stats(ctypes.byref(d)):
        partials = torch.empty((N // 256, M, 2), dtype=torch.float32, device=a.device)
        d.row_stats = ptr(partials)
    check(lib.me_gemm(ctypes.byref(d), stream_ptr()), "me_gemm")
    if want_row_stats:
        return out, partials
    if want_colsum_a:
        if cs is None:      # not fusable for this problem: separate pass over a
            cs = colsum(a2, out=colsum_out, accumulate=beta != 0.0)
        return out, cs
    return out


GG8_LO, GG8_STEP = -0.13, 1.26 / 255.0      # ME_GG8 (include/metaenc.h)


def gemm_takes_gg8(M: int, N: int, K: int) -> bool:
    """Do the two flagged GEMMs of a bf16 training MLP -- [M, K] x [N, K]^T with gelu' saved / multiplied -- take it in eight bits
    (me_gemm_takes_gg8: both on the resident kernel)?  Needs a GPU (the answer depends on its CU count)."""
    lib = _capi.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    dummy = torch.empty(64, dtype=torch.uint8, device=dev)
    for save in (True, False):
        d = GemmDesc()
        d.op, d.ab_dtype, d.c_dtype = _capi.ME_GEMM_NT, _capi.ME_BF16, _capi.ME_BF16
        d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, K, K, N
        d.A = d.B = d.C = ptr(dummy)
        d.alpha = 1.0
        if save:
            d.act, d.flags, d.preact, d.ldpre, d.preact_dtype = _capi.ME_ACT_GELU, _capi.ME_GEMM_SAVE_GELU_GRAD, ptr(dummy), N, _capi.ME_GG8
        else:
            d.flags, d.aux, d.ldaux, d.aux_dtype = _capi.ME_GEMM_AUX_IS_FACTOR, ptr(dummy), N, _capi.ME_GG8
        d.workspace, d.workspace_bytes = ptr(dummy), 1 << 40
        if not lib.me_gemm_takes_gg8(ctypes.byref(d)):
            return False
    return True


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    lib = _capi.load()
    _req(x, "x")
    x2 = x.reshape(-1, x.shape[-1])
    rows, cols = x2.shape
    if out is None:
        out, accumulate = torch.empty(cols, dtype=torch.float32, device=x.device), False
    ws = torch.empty(lib.me_colsum_workspace(cols), dtype=torch.uint8, device=x.device)
    check(lib.me_colsum(ptr(x2), dtype_code(x.dtype), x2.stride(0), rows, cols, ptr(out), 1 if accumulate else 0, ptr(ws),
                        stream_ptr()), "me_colsum")
    return out


# ----------------------------------------------------------------------------- whole block (C-side composition)

def block_desc(B, N, C, heads, hidden, eps, scale, cdt, rdt, w, wt, vec, x3: bool = False) -> "_capi.BlockDesc":
    """w / wt: dicts name -> tensor (compute dtype) for qkv, proj, fc1, fc2 (wt may be None); vec: dict of fp32 vectors
    ln1_g, ln1_b, ln2_g, ln2_b, qkv_b, proj_b, fc1_b, fc2_b, gamma1, gamma2 (None = absent).  The caller keeps the
    tensors alive for the duration of the call."""
    d = _capi.BlockDesc()
    # x3: fp32-accurate arithmetic on the bf16 matrix pipe (me_block_desc.dtype = ME_BF16X3; the weights are ops.split3 right operands)
    d.dtype, d.res_dtype = (_capi.ME_BF16X3 if x3 else dtype_code(cdt)), dtype_code(rdt)
    d.B, d.N, d.C, d.heads, d.hidden, d.eps, d.scale = B, N, C, heads, hidden, eps, scale
    for k in ("qkv", "proj", "fc1", "fc2"):
        setattr(d, k + "_w", ptr(w[k]))
        setattr(d, k + "_wt", ptr(wt[k]) if wt is not None else None)
    for k, t in vec.items():
        setattr(d, k, ptr(t))
    return d


def block_fwd(d, x2: torch.Tensor, keep: bool, x_stats: Optional[torch.Tensor] = None, want_stats: bool = False):
    """-> (y, saved or None, y_stats or None).  x_stats: the LayerNorm statistics of x for this block's norm1 (folded inference): the
    [M, 2] pairs of row_stats (me_block_desc.x_stats) or the [C / 256, M, 2] partials a previous call returned (me_block_desc.x_parts);
    want_stats: also return the partials of y, left by the fc2 epilogue, when the block can emit them (me_block_emits_stats) -- the
    next block's x_stats.  No statistics pass and no combine launch sits between the blocks then."""
    lib = _capi.load()
    y = torch.empty_like(x2)
    saved = torch.empty(lib.me_block_saved_bytes(ctypes.byref(d)), dtype=torch.uint8, device=x2.device) if keep else None
    wsb = lib.me_block_workspace_bytes(ctypes.byref(d), 0)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x2.device)
    y_stats = None
    if not keep:
        if x_stats is not None:
            M, C = x2.shape
            if x_stats.dtype != torch.float32 or not x_stats.is_contiguous() or x_stats.device != x2.device or \
                    tuple(x_stats.shape) not in ((M, 2), (C // 256, M, 2)):
                raise MetaEncError("block_fwd: x_stats must be a contiguous float32 tensor on the tokens' device, [M, 2] pairs "
                                   "(row_stats) or the [C / 256, M, 2] partials a previous block_fwd(want_stats=True) returned")
            if x_stats.dim() == 3:
                d.x_parts = ptr(x_stats)       # the fc2 partials themselves: this block's qkv GEMM forms the pairs (no combine launch)
            else:
                d.x_stats = ptr(x_stats)
        if want_stats and lib.me_block_emits_stats(ctypes.byref(d)):
            y_stats = torch.empty((x2.shape[1] // 256, x2.shape[0], 2), dtype=torch.float32, device=x2.device)
            d.y_parts = ptr(y_stats)
    check(lib.me_block_fwd(ctypes.byref(d), ptr(x2), ptr(y), ptr(saved), ptr(ws), wsb, stream_ptr()), "me_block_fwd")
    return y, saved, y_stats


def encoder_fwd(descs, x2: torch.Tensor) -> torch.Tensor:
    """inference through a list of block descriptors in ONE library call (me_encoder_fwd)"""
    lib = _capi.load()
    n = len(descs)
    arr = (_capi.BlockDesc * n)(*descs)
    y = torch.empty_like(x2)
    pp = torch.empty_like(x2) if n > 1 else None
    wsb = max(lib.me_block_workspace_bytes(ctypes.byref(d), 0) for d in descs)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x2.device)
    check(lib.me_encoder_fwd(arr, n, ptr(x2), ptr(y), ptr(pp), ptr(ws), wsb, stream_ptr()), "me_encoder_fwd")
    return y


def attention_fwd_x3(qkv: torch.Tensor, B: int, N: int, H: int, hd: int, scale: float, need_lse: bool = False, planes: bool = False,
                     want_out: bool = True):
    """fp32-accurate attention forward on the bf16 matrix pipe (me_attention_fwd_x3): fp32 qkv [B*N, 3*H*hd] -> (out fp32 [B*N, H*hd],
    lse or None, ME_BF16X3 planes [B*N, 3*H*hd] bf16 or None)."""
    lib = _capi.load()
    _req(qkv, "qkv")
    if qkv.dtype != torch.float32:
        raise MetaEncError("attention_fwd_x3: float32 qkv required")
    C = H * hd
    out = torch.empty((B * N, C), dtype=torch.float32, device=qkv.device) if want_out else None
    lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device) if need_lse else None
    o3 = torch.empty((B * N, 3 * C), dtype=torch.bfloat16, device=qkv.device) if planes else None
    check(lib.me_attention_fwd_x3(ptr(qkv), qkv.stride(-2), ptr(out), C, ptr(o3), ptr(lse), B, N, H, hd, float(scale), stream_ptr()),
          "me_attention_fwd_x3")
    return out, lse, o3


def attention_bwd_x3(qkv: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, lse: torch.Tensor, B: int, N: int, H: int, hd: int,
                     scale: float, planes: bool = False):
    """backward of attention_fwd_x3 (me_attention_bwd_x3): fp32 tensors, three bf16 products per operand pair -> dqkv [B*N, 3*H*hd]
    (planes=True: -> (dqkv, its ME_BF16X3 planes [B*N, 9*H*hd] bf16))"""
    lib = _capi.load()
    for t, n in ((qkv, "qkv"), (out, "out"), (dout, "dout"), (lse, "lse")):
        _req(t, n)
        if t.dtype != torch.float32:
            raise MetaEncError(f"attention_bwd_x3: {n} must be float32")
    C = H * hd
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    d3 = torch.empty((B * N, 9 * C), dtype=torch.bfloat16, device=qkv.device) if planes else None
    check(lib.me_attention_bwd_x3(ptr(qkv), qkv.stride(-2), ptr(out), C, ptr(dout), C, ptr(lse), ptr(delta), ptr(dqkv), dqkv.stride(-2),
                                  ptr(d3), B, N, H, hd, float(scale), stream_ptr()), "me_attention_bwd_x3")
    return (dqkv, d3) if planes else dqkv


def block_bwd_overlap(enable: bool) -> bool:
    """Switch the side stream of me_block_bwd (weight-gradient GEMMs beside the dY -> dX chain) on / off; returns the previous setting."""
    return bool(_capi.load().me_block_bwd_overlap(1 if enable else 0))


def block_bwd(d, x2: torch.Tensor, dy2: torch.Tensor, saved: torch.Tensor, grads: "_capi.BlockGrads") -> torch.Tensor:
    lib = _capi.load()
    dx = torch.empty_like(x2)
    wsb = lib.me_block_workspace_bytes(ctypes.byref(d), 1)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x2.device)
    check(lib.me_block_bwd(ctypes.byref(d), ptr(x2), ptr(dy2), ptr(saved), ptr(dx), ctypes.byref(grads), ptr(ws), wsb,
                           stream_ptr()), "me_block_bwd")
    return dx


# ----------------------------------------------------------------------------- attention

def attention_fwd(qkv: torch.Tensor, B: int, N: int, H: int, hd: int, scale: float, need_lse: bool,
                  p_drop: float = 0.0, seed: int = 0, fp8: bool = False):
    """p_drop > 0: dropout on the attention probabilities (training-mode attn_drop); same p_drop / seed in attention_bwd.
    fp8: e4m3 Q / K / V / P on the block-scaled MFMA (me_attention_fwd_fp8: bf16 qkv, head_dim 64, no dropout)."""
    lib = _capi.load()
    _req(qkv, "qkv")
    C = H * hd
    out = torch.empty((B * N, C), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device) if need_lse else None
    if fp8:
        if qkv.dtype != torch.bfloat16 or hd != 64 or p_drop > 0:
            raise MetaEncError("fp8 attention: bf16 qkv, head_dim 64 and no attention dropout are required")
        nbytes = lib.me_attention_fp8_workspace(B, N, H, hd)
        # scratch for the quantised Q / K / V^T: taken from torch's caching allocator per call -- a block freed here goes back to
        # the pool of the stream it was allocated on, so a second stream (or a graph capture) never sees it while this call's
        # kernels may still be running (a module-level buffer shared by every stream did not have that property)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=qkv.device)
        check(lib.me_attention_fwd_fp8(ptr(qkv), 3 * C, ptr(out), C, ptr(lse), B, N, H, hd, float(scale), ptr(ws), nbytes,
                                       stream_ptr()), "me_attention_fwd_fp8")
        return out, lse
    check(lib.me_attention_fwd(ptr(qkv), 3 * C, ptr(out), C, ptr(lse), B, N, H, hd, float(scale),
                               dtype_code(qkv.dtype), float(p_drop), seed & 0xFFFFFFFFFFFFFFFF, stream_ptr()), "me_attention_fwd")
    return out, lse


def attention_bwd(qkv: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, lse: torch.Tensor,
                  B: int, N: int, H: int, hd: int, scale: float, p_drop: float = 0.0, seed: int = 0) -> torch.Tensor:
    lib = _capi.load()
    _req(qkv, "qkv"); _req(out, "out"); _req(dout, "dout")
    C = H * hd
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    check(lib.me_attention_bwd(ptr(qkv), 3 * C, ptr(out), C, ptr(dout), C, ptr(lse), ptr(delta), ptr(dqkv), 3 * C,
                               B, N, H, hd, float(scale), dtype_code(qkv.dtype), float(p_drop), seed & 0xFFFFFFFFFFFFFFFF,
                               stream_ptr()), "me_attention_bwd")
    return dqkv


# ----------------------------------------------------------------------------- element-wise

def cast(x: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if x.dtype == dtype and out is None:
        return x
    lib = _capi.load()
    _req(x, "x")
    if out is not None and (out.dtype != dtype or out.numel() != x.numel() or not out.is_contiguous() or out.device != x.device):
        raise MetaEncError("cast: `out` must be a contiguous tensor of the target dtype with as many elements as x")
    y = out if out is not None else torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib.me_cast(ptr(x), dtype_code(x.dtype, True), ptr(y), dtype_code(dtype, True), x.numel(), stream_ptr()), "me_cast")
    return y


def split3(x: torch.Tensor, right_operand: bool = False) -> torch.Tensor:
    """fp32 [rows, cols] -> ME_BF16X3: bf16 [rows, 3 * cols] = [hi | lo | hi] (left operand: activations, gradients) or
    [hi | hi | lo] (right operand: weights), hi = bf16(x), lo = bf16(x - hi).  A bf16 NT GEMM over two such operands
    (K = 3 * cols) computes x @ w^T to ~2^-17 relative on the bf16 matrix pipe (me_split3)."""
    lib = _capi.load()
    _req(x, "x")
    if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] % 4:
        raise MetaEncError("split3: a contiguous 2-D float32 tensor with cols % 4 == 0 is required")
    rows, cols = x.shape
    y = torch.empty((rows, 3 * cols), dtype=torch.bfloat16, device=x.device)
    check(lib.me_split3(ptr(x), cols, ptr(y), rows, cols, 1 if right_operand else 0, stream_ptr()), "me_split3")
    return y


def transpose_cast(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[rows, cols] -> [cols, rows] in `dtype` (weight repack for dgrad)."""
    lib = _capi.load()
    _req(w, "w")
    rows, cols = w.shape
    y = torch.empty((cols, rows), dtype=dtype, device=w.device)
    check(lib.me_transpose_cast(ptr(w), dtype_code(w.dtype, True), ptr(y), dtype_code(dtype, True), rows, cols, stream_ptr()),
          "me_transpose_cast")
    return y


def transpose_cast_many(ws, dtype: torch.dtype):
    """[rows, cols] -> [cols, rows] in `dtype` for a list of same-dtype matrices, ME_TC_BATCH per launch"""
    lib = _capi.load()
    outs = []
    for i in range(0, len(ws), _capi.ME_TC_BATCH):
        chunk = ws[i:i + _capi.ME_TC_BATCH]
        b = _capi.TcBatch()
        b.n, b.src_dtype, b.dst_dtype = len(chunk), dtype_code(chunk[0].dtype, True), dtype_code(dtype, True)
        for k, w in enumerate(chunk):
            _req(w, "w")
            if w.dtype != chunk[0].dtype:
                raise MetaEncError("transpose_cast_many: matrices must share a dtype")
            rows, cols = w.shape
            y = torch.empty((cols, rows), dtype=dtype, device=w.device)
            outs.append(y)
            b.item[k].src, b.item[k].dst, b.item[k].rows, b.item[k].cols = ptr(w), ptr(y), rows, cols
        check(lib.me_transpose_cast_batched(ctypes.byref(b), stream_ptr()), "me_transpose_cast_batched")
    return outs


def split3_many(ws, transposed: bool, right_operand: bool = True):
    """ME_BF16X3 planes of a list of fp32 matrices [rows, cols] -- of each matrix ([rows, 3 cols]) or of its transpose ([cols, 3 rows]) --
    ME_TC_BATCH per launch (me_split3_batched): the fp32-accurate mode's compute copies of every weight after an optimizer step."""
    lib = _capi.load()
    outs = []
    for i in range(0, len(ws), _capi.ME_TC_BATCH):
        chunk = ws[i:i + _capi.ME_TC_BATCH]
        b = _capi.TcBatch()
        b.n, b.src_dtype, b.dst_dtype = len(chunk), _capi.ME_F32, _capi.ME_BF16X3
        for k, w in enumerate(chunk):
            _req(w, "w")
            if w.dtype != torch.float32 or w.dim() != 2 or not w.is_contiguous():
                raise MetaEncError("split3_many: contiguous fp32 matrices")
            rows, cols = w.shape
            y = torch.empty((cols, 3 * rows) if transposed else (rows, 3 * cols), dtype=torch.bfloat16, device=w.device)
            outs.append(y)
            b.item[k].src, b.item[k].dst, b.item[k].rows, b.item[k].cols = ptr(w), ptr(y), rows, cols
        check(lib.me_split3_batched(ctypes.byref(b), 1 if transposed else 0, 1 if right_operand else 0, stream_ptr()), "me_split3_batched")
    return outs


def add_rows(x: torch.Tensor, pos: torch.Tensor, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """x[B*N, C] + pos[(row % pos_rows), C]"""
    lib = _capi.load()
    _req(x, "x"); _req(pos, "pos")
    C = x.shape[-1]
    rows = x.numel() // C
    pos_rows = pos.numel() // C
    y = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    check(lib.me_add_rows(ptr(x), dtype_code(x.dtype), ptr(pos), dtype_code(pos.dtype), ptr(y), dtype_code(y.dtype),
                          rows, pos_rows, C, stream_ptr()), "me_add_rows")
    return y


def resize_rows(table: torch.Tensor, hw, HW, mode: str = "bicubic") -> torch.Tensor:
    """[h*w, C] grid table -> [H*W, C], F.interpolate(mode, align_corners=False) semantics (me_resize_rows)."""
    lib = _capi.load()
    _req(table, "table")
    (h, w), (H, W) = hw, HW
    C = table.shape[-1]
    if table.numel() != h * w * C:
        raise MetaEncError(f"resize_rows: table has {table.numel() // C} rows, expected {h}x{w}")
    if mode not in ("bicubic", "bilinear"):
        raise MetaEncError(f"resize_rows: mode {mode!r} (bicubic / bilinear are implemented)")
    out = torch.empty((H * W, C), dtype=table.dtype, device=table.device)
    check(lib.me_resize_rows(ptr(table), dtype_code(table.dtype), ptr(out), dtype_code(out.dtype), h, w, H, W, C,
                             _capi.ME_RESIZE_BICUBIC if mode == "bicubic" else _capi.ME_RESIZE_BILINEAR, stream_ptr()),
          "me_resize_rows")
    return out


def dropout_add(v: torch.Tensor, res: Optional[torch.Tensor], rows_per_sample: int, p_drop: float, p_path: float,
                seed: int, out_dtype: Optional[torch.dtype] = None, colscale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """res + colscale * drop_path(dropout(v)); with res=None and v = incoming gradient this is the backward (same seed).
    p_drop = p_path = 0 makes it the plain fused `res + colscale * v`."""
    lib = _capi.load()
    _req(v, "v")
    C = v.shape[-1]
    rows = v.numel() // C
    out = torch.empty(v.shape, dtype=out_dtype or (res.dtype if res is not None else v.dtype), device=v.device)
    cs = _f32(colscale).contiguous() if colscale is not None else None
    check(lib.me_dropout_add(ptr(v), dtype_code(v.dtype), ptr(res), dtype_code(res.dtype) if res is not None else 0,
                             ptr(out), dtype_code(out.dtype), rows, C, rows_per_sample, float(p_drop), float(p_path),
                             seed & 0xFFFFFFFFFFFFFFFF, ptr(cs), stream_ptr()), "me_dropout_add")
    return out


def window_rows(x: torch.Tensor, B: int, H: int, W: int, window: int, merge: bool) -> torch.Tensor:
    """partition ([B*H*W, C] -> [B*gh*gw*ws*ws, C], zero rows outside the grid) or merge (inverse) -- me_window_rows"""
    lib = _capi.load()
    _req(x, "x")
    C = x.shape[-1]
    gh, gw = -(-H // window), -(-W // window)
    rows_tok, rows_win = B * H * W, B * gh * gw * window * window
    x2 = x.reshape(-1, C)
    if x2.shape[0] != (rows_win if merge else rows_tok):
        raise MetaEncError(f"window_rows: {x2.shape[0]} rows, expected {rows_win if merge else rows_tok}")
    out = torch.empty((rows_tok if merge else rows_win, C), dtype=x.dtype, device=x.device)
    check(lib.me_window_rows(ptr(x2), ptr(out), dtype_code(x.dtype), B, H, W, window, C, 1 if merge else 0, stream_ptr()),
          "me_window_rows")
    return out


def colsum_mul(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """out[c] = sum_r x[r,c] * y[r,c]  (fp32)"""
    lib = _capi.load()
    _req(x, "x"); _req(y, "y")
    x2, y2 = x.reshape(-1, x.shape[-1]), y.reshape(-1, y.shape[-1])
    if x2.shape != y2.shape:
        raise MetaEncError(f"colsum_mul: shapes differ ({tuple(x2.shape)} vs {tuple(y2.shape)})")
    rows, cols = x2.shape
    out = torch.empty(cols, dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.me_colsum_workspace(cols), dtype=torch.uint8, device=x.device)
    check(lib.me_colsum_mul(ptr(x2), dtype_code(x.dtype), x2.stride(0), ptr(y2), dtype_code(y.dtype), y2.stride(0), rows, cols,
                            ptr(out), 0, ptr(ws), stream_ptr()), "me_colsum_mul")
    return out


def patchify(x: torch.Tensor, kt: int, kh: int, kw: int, st: int, sh: int, sw: int, out_dtype: torch.dtype):
    """x [B,Cin,(T,)H,W] -> ([B*tokens, Cin*kt*kh*kw], tokens_per_sample)"""
    lib = _capi.load()
    _req(x, "x")
    if x.dim() == 4:
        B, Cin, H, W = x.shape
        T = 1
    else:
        B, Cin, T, H, W = x.shape
    gt, gh, gw = (T - kt) // st + 1, (H - kh) // sh + 1, (W - kw) // sw + 1
    cols = torch.empty((B * gt * gh * gw, Cin * kt * kh * kw), dtype=out_dtype, device=x.device)
    check(lib.me_patchify(ptr(x), dtype_code(x.dtype), ptr(cols), dtype_code(out_dtype), B, Cin, T, H, W,
                          kt, kh, kw, st, sh, sw, stream_ptr()), "me_patchify")
    return cols, gt * gh * gw


def _patch_embed_desc(x: torch.Tensor, geom, w_dtype: torch.dtype, Cout: int):
    kt, kh, kw, st, sh, sw = geom
    if x.dim() == 4:
        B, Cin, H, W = x.shape
        T = 1
    else:
        B, Cin, T, H, W = x.shape
    d = _capi.PatchEmbedDesc()
    d.x, d.x_dtype = ptr(x), dtype_code(x.dtype)
    d.B, d.Cin, d.T, d.H, d.W = B, Cin, T, H, W
    d.kt, d.kh, d.kw, d.st, d.sh, d.sw = kt, kh, kw, st, sh, sw
    d.w_dtype, d.Cout = dtype_code(w_dtype), Cout
    tokens = ((T - kt) // st + 1) * ((H - kh) // sh + 1) * ((W - kw) // sw + 1)
    return d, B, tokens


def patch_embed_fused(x: torch.Tensor, geom, w_dtype: torch.dtype, Cout: int, x_dtype: Optional[torch.dtype] = None) -> bool:
    """Will me_patch_embed gather this input (as it is, or once cast to x_dtype) inside the GEMM's operand stager -- no gathered
    matrix, no workspace?"""
    lib = _capi.load()
    d, _, _ = _patch_embed_desc(x, geom, w_dtype, Cout)
    if x_dtype is not None:
        d.x_dtype = dtype_code(x_dtype)
    return bool(lib.me_patch_embed_fused(ctypes.byref(d)))


def patch_embed_wgrad_fused(x: torch.Tensor, geom, compute_dtype: torch.dtype, Cout: int, dw_dtype: torch.dtype) -> bool:
    """Will me_patch_embed_wgrad gather the patches inside the weight-gradient kernel?"""
    lib = _capi.load()
    d, _, _ = _patch_embed_desc(x, geom, compute_dtype, Cout)
    return bool(lib.me_patch_embed_wgrad_fused(ctypes.byref(d), dtype_code(dw_dtype)))


def patch_embed(x: torch.Tensor, w2: torch.Tensor, bias: Optional[torch.Tensor], pos: Optional[torch.Tensor], geom,
                prefix_rows: int = 0, out_dtype: Optional[torch.dtype] = None):
    """Conv patch embed as one call (me_patch_embed): x [B,Cin,(T,)H,W], w2 [Cout, Cin*kt*kh*kw] in the compute dtype ->
    ([B*(prefix_rows+tokens), Cout], tokens).  Rows of the prefix are zero."""
    lib = _capi.load()
    _req(x, "x"); _req(w2, "weight")
    if not x.is_contiguous() or not w2.is_contiguous():
        raise MetaEncError("patch_embed: x and weight must be contiguous")
    Cout = w2.shape[0]
    d, B, tokens = _patch_embed_desc(x, geom, w2.dtype, Cout)
    if w2.shape[1] != d.Cin * d.kt * d.kh * d.kw:
        raise MetaEncError(f"patch_embed: weight has {w2.shape[1]} columns, the patch has {d.Cin * d.kt * d.kh * d.kw} features")
    odt = out_dtype or w2.dtype
    rows = B * (tokens + prefix_rows)
    y = (torch.zeros if prefix_rows else torch.empty)((rows, Cout), dtype=odt, device=x.device)
    d.weight = ptr(w2)
    if bias is not None:
        bias = bias.detach().float().contiguous()
        d.bias = ptr(bias)
    if pos is not None:
        if pos.shape != (tokens, Cout):
            raise MetaEncError(f"pos-embed has shape {tuple(pos.shape)}, tokenizer produces [{tokens}, {Cout}]")
        pos = pos.contiguous()
        d.pos, d.pos_dtype, d.ld_pos = ptr(pos), dtype_code(pos.dtype), pos.stride(0)
    d.prefix_rows = prefix_rows
    d.out, d.out_dtype, d.ld_out = ptr(y), dtype_code(odt), y.stride(0)
    nws = lib.me_patch_embed_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=x.device)
    d.workspace, d.workspace_bytes = ptr(ws), nws
    check(lib.me_patch_embed(ctypes.byref(d), stream_ptr()), "me_patch_embed")
    return y, tokens


def patch_embed_wgrad(x: torch.Tensor, geom, dy2: torch.Tensor, dw_dtype: torch.dtype, want_bias: bool):
    """(dW [Cout, Cin*kt*kh*kw], dbias [Cout] fp32 or None) of the patch embed from dY [B*tokens, Cout] (me_patch_embed_wgrad)."""
    lib = _capi.load()
    _req(x, "x"); _req(dy2, "dy")
    if not x.is_contiguous() or dy2.stride(1) != 1:
        raise MetaEncError("patch_embed_wgrad: x must be contiguous, dy row-major")
    Cout = dy2.shape[1]
    d, B, tokens = _patch_embed_desc(x, geom, dy2.dtype, Cout)
    if dy2.shape[0] != B * tokens:
        raise MetaEncError(f"patch_embed_wgrad: dy has {dy2.shape[0]} rows, the input makes {B * tokens} tokens")
    K = d.Cin * d.kt * d.kh * d.kw
    dw = torch.empty((Cout, K), dtype=dw_dtype, device=x.device)
    db = torch.empty(Cout, dtype=torch.float32, device=x.device) if want_bias else None
    nws = lib.me_patch_embed_wgrad_workspace_bytes(ctypes.byref(d), dtype_code(dw_dtype), 1 if want_bias else 0)
    ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=x.device)
    d.workspace, d.workspace_bytes = ptr(ws), nws
    check(lib.me_patch_embed_wgrad(ctypes.byref(d), ptr(dy2), dy2.stride(0), ptr(dw), dtype_code(dw_dtype), ptr(db), 0.0, stream_ptr()),
          "me_patch_embed_wgrad")
    return dw, db


def unpatchify_add(dcols: torch.Tensor, x_shape, kt, kh, kw, st, sh, sw) -> torch.Tensor:
    lib = _capi.load()
    _req(dcols, "dcols")
    if len(x_shape) == 4:
        B, Cin, H, W = x_shape
        T = 1
    else:
        B, Cin, T, H, W = x_shape
    dx = torch.zeros(x_shape, dtype=torch.float32, device=dcols.device)
    check(lib.me_unpatchify_add(ptr(dcols), dtype_code(dcols.dtype), ptr(dx), B, Cin, T, H, W, kt, kh, kw, st, sh, sw,
                                stream_ptr()), "me_unpatchify_add")
    return dx


def adamw_step(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, *, lr: float,
               betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01, step: int = 1,
               grad_scale: float = 1.0, bf16_mirror: Optional[torch.Tensor] = None, bump_epoch: bool = True) -> None:
    """Raw fused AdamW on flat fp32 buffers (me_adamw_step).  The kernel writes `param` through its raw pointer, so
    torch's tensor version counters do not move; Block weight-copy caches notice the update through WEIGHT_EPOCH, which
    they consult for parameters registered in a parallel.FlatParams -- use parallel.FusedAdamW rather than this call
    unless you manage cache invalidation yourself."""
    lib = _capi.load()
    if bf16_mirror is not None and (bf16_mirror.dtype != torch.bfloat16 or bf16_mirror.numel() != param.numel()):
        raise MetaEncError("adamw_step: bf16_mirror must be a bfloat16 tensor of the parameter buffer's size")
    for t, n in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(t, n)
        if t.dtype != torch.float32:
            raise MetaEncError(f"adamw_step: {n} must be float32")
    check(lib.me_adamw_step(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), lr, betas[0], betas[1],
                            eps, weight_decay, step, grad_scale, ptr(bf16_mirror), stream_ptr()), "me_adamw_step")
    if bump_epoch:        # (False: a caller that steps slice by slice and announces the new weights once, with weights_updated())
        weights_updated()


def weights_updated() -> None:
    """Announce that parameters registered in a parallel.FlatParams were written through raw pointers (Block weight caches re-derive
    their compute copies)."""
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def grad_stats(grad: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """{L2 norm of the finite values, number of non-finite values} of a flat fp32 gradient buffer as a 2-element DEVICE
    tensor (me_grad_stats: deterministic two-level sum of squares in double; nothing is synchronised)."""
    _req(grad, "grad")
    if grad.dtype != torch.float32 or grad.dim() != 1:
        raise MetaEncError("grad_stats: a flat float32 buffer is required")
    lib = _capi.load()
    if out is None:
        out = torch.empty(2, dtype=torch.float32, device=grad.device)
    ws = torch.empty(lib.me_grad_stats_workspace(), dtype=torch.uint8, device=grad.device)
    check(lib.me_grad_stats(ptr(grad), grad.numel(), ptr(out), ptr(ws), stream_ptr()), "me_grad_stats")
    return out


def adamw_step_segments(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
                        segments: torch.Tensor, n_segments: int, ctl: torch.Tensor, *, lr: float, betas=(0.9, 0.999),
                        eps: float = 1e-8, grad_scale: float = 1.0, stats: Optional[torch.Tensor] = None,
                        loss_scale: Optional[torch.Tensor] = None, max_norm: float = 0.0,
                        bf16_mirror: Optional[torch.Tensor] = None) -> None:
    """me_adamw_prepare + me_adamw_step_segments on flat fp32 buffers (see include/metaenc.h): per-segment lr scale / weight
    decay, gradient unscale, global-norm clipping and the found-inf skip, all decided on the device.  `segments` is a uint8
    device tensor holding n_segments me_adamw_segment records, `ctl` a 32-byte device control block (zeroed once by the caller;
    its step counter lives there).  Same cache-invalidation contract as adamw_step."""
    lib = _capi.load()
    for t, n in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(t, n)
        if t.dtype != torch.float32:
            raise MetaEncError(f"adamw_step_segments: {n} must be float32")
    if bf16_mirror is not None and (bf16_mirror.dtype != torch.bfloat16 or bf16_mirror.numel() != param.numel()):
        raise MetaEncError("adamw_step_segments: bf16_mirror must be a bfloat16 tensor of the parameter buffer's size")
    if ctl.numel() * ctl.element_size() < ctypes_sizeof_ctl() or not ctl.is_cuda:
        raise MetaEncError("adamw_step_segments: ctl must be a device buffer of sizeof(me_adamw_ctl) bytes")
    check(lib.me_adamw_prepare(ptr(ctl), ptr(stats), ptr(loss_scale), grad_scale, max_norm, betas[0], betas[1], stream_ptr()),
          "me_adamw_prepare")
    check(lib.me_adamw_step_segments(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), ptr(segments), n_segments,
                                     lr, betas[0], betas[1], eps, ptr(ctl), ptr(bf16_mirror), stream_ptr()), "me_adamw_step_segments")
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def ctypes_sizeof_ctl() -> int:
    import ctypes
    return ctypes.sizeof(_capi.AdamwCtl)


def row_stats_combine(partials: torch.Tensor, eps: float) -> torch.Tensor:
    """[C / 256, rows, 2] partial statistics (gemm(..., want_row_stats=True)) -> [rows, 2] pairs (rstd, -rstd * mean) of
    LayerNorm(C, eps): the same pairs as row_stats(x, eps) on the tensor the GEMM wrote (me_row_stats_combine)."""
    _req(partials, "partials")
    if partials.dtype != torch.float32 or partials.dim() != 3 or partials.shape[2] != 2:
        raise MetaEncError("row_stats_combine: [C / 256, rows, 2] float32 partials required")
    nparts, rows, _ = partials.shape
    out = torch.empty(rows, 2, dtype=torch.float32, device=partials.device)
    check(_capi.load().me_row_stats_combine(ptr(partials), rows, nparts * 256, float(eps), ptr(out), stream_ptr()), "me_row_stats_combine")
    return out


def row_stats(x: torch.Tensor, eps: float) -> torch.Tensor:
    """[rows, C] -> [rows, 2] fp32 pairs (rstd, -rstd * mean): the statistics of LayerNorm(C, eps) for a Linear that has the
    normalisation folded in (me_row_stats; gemm(..., row_affine=, col_shift=))."""
    _req(x, "x")
    C = x.shape[-1]
    rows = x.numel() // C
    out = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
    check(_capi.load().me_row_stats(ptr(x), dtype_code(x.dtype), ptr(out), rows, C, float(eps), stream_ptr()), "me_row_stats")
    return out
