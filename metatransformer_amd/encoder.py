"""Drop-in replacement for the reference's encoder plugin point.

Reference usage being replaced (README.md:113-150; every call site in SURVEY.md 2.2):

    from timm.models.vision_transformer import Block
    encoder = nn.Sequential(*[Block(dim=768, num_heads=12, mlp_ratio=4., qkv_bias=True,
                                    norm_layer=nn.LayerNorm, act_layer=nn.GELU) for _ in range(12)])
    encoder.load_state_dict(torch.load("Meta-Transformer_base_patch16_encoder.pth"), strict=True)
    y = encoder(x)                                # [B, N, C] -> [B, N, C]

With this package only the import changes:

    from metatransformer_amd import Block

``Block`` keeps timm's constructor signature, owns ordinary ``nn.Parameter`` tensors under the reference
names (``norm1.weight``, ``attn.qkv.weight``, ``mlp.fc1.bias`` ...; in-tree twin of the timm class:
PointCloud/openpoints/models/layers/attention.py:12-58, mlp.py:11-35) so ``state_dict()`` round-trips with
the reference checkpoint, works inside ``nn.Sequential`` / slicing / DataParallel / DDP / checkpointing,
and dispatches forward AND backward to the hand-written HIP kernels of libmetaenc.so through one
``torch.autograd.Function`` per block.  There is no PyTorch-op fallback: CPU tensors raise.

Compute dtype: bf16 when the parameters are bf16 or when called under ``torch.autocast(dtype=bfloat16)``
(fp32 master weights, bf16 MFMA, fp32 accumulation / statistics / residual stream), otherwise exact fp32 MFMA.
"""
from __future__ import annotations

from typing import Optional
import warnings

import torch
import torch.nn as nn

from . import _capi, ops
from ._capi import ME_ACT_GELU, ME_GEMM_AUX_IS_FACTOR, ME_GEMM_SAVE_GELU_GRAD, ME_GEMM_TN, MetaEncError


def _tensor_version(t: torch.Tensor):
    """``t._version``, or None for inference tensors (``torch.inference_mode()``: they carry no version counter and reading
    the attribute raises)."""
    try:
        return t._version
    except RuntimeError:
        return None


def _resolve_eps(norm_layer) -> float:
    """timm passes a class or a functools.partial(nn.LayerNorm, eps=1e-6) (1e-5 vs 1e-6 sites)."""
    probe = norm_layer(8)
    if not isinstance(probe, nn.LayerNorm):
        raise MetaEncError(f"norm_layer must build an nn.LayerNorm (got {type(probe).__name__})")
    return float(probe.eps)


class Mlp(nn.Module):
    """Parameter container with the reference names (mlp.py:15-27): fc1 -> GELU -> fc2."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if act_layer is not nn.GELU:
            raise MetaEncError("only act_layer=nn.GELU (exact erf) is implemented, as every reference call site uses")
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class Attention(nn.Module):
    """Parameter container with the reference names (attention.py:13-24)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        if dim % num_heads:
            raise MetaEncError(f"dim {dim} not divisible by num_heads {num_heads}")
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)


class _WeightCache:
    """Compute-dtype copies of the four weight matrices of a block ([out,in] for forward, transposed
    [in,out] for dgrad), rebuilt when the parameter changes (optimizer step / load_state_dict / .to()).

    Two shortcuts keep the per-step refresh off the critical path: forward bf16 copies come straight out of the fused
    optimizer's bf16 mirror when the parameters live in a parallel.FlatParams, and a stale TRANSPOSED copy triggers one
    batched transpose of every stale weight of every live Block on the device (one launch instead of four per block)."""

    _live = None          # weakref.WeakSet of all caches (set up lazily)

    def __init__(self):
        import weakref
        self._fwd = {}
        self._tr = {}
        self._fold = {}
        self._x3 = {}
        self._owner = None          # weakref to the Block, set by Block.__init__
        if _WeightCache._live is None:
            _WeightCache._live = weakref.WeakSet()
        _WeightCache._live.add(self)

    # a cache is derived state: copies / pickles of a Block start with an empty one (re-bound to the new owner on first use)
    def __deepcopy__(self, memo):
        return _WeightCache()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.__init__()

    def bind(self, blk) -> None:
        if self._owner is None or self._owner() is not blk:
            import weakref
            self._owner = weakref.ref(blk)

    @staticmethod
    def _key(p: torch.Tensor, dtype):
        # WEIGHT_EPOCH covers the fused optimizer, which writes parameters through raw pointers (no _version bump); it
        # only ever touches parameters that live in a parallel.FlatParams, so other (e.g. frozen) encoders keep their copies
        epoch = ops.WEIGHT_EPOCH if hasattr(p, "_me_flat") else 0
        return (p.data_ptr(), _tensor_version(p), epoch, p.dtype, dtype, p.device)

    def fwd(self, name: str, p: torch.Tensor, dtype) -> torch.Tensor:
        if p.dtype == dtype:
            return p.detach()
        if dtype == torch.bfloat16:
            flat = getattr(p, "_me_flat", None)
            if flat is not None:
                v = flat.bf16_view(p)
                if v is not None:
                    return v
        k = self._key(p, dtype)
        slot = (name, p.device)       # nn.DataParallel replicas share this object (shallow __dict__ copy): one slot per device
        hit = self._fwd.get(slot)
        if hit is None or hit[0] != k:
            hit = (k, ops.cast(p.detach().contiguous(), dtype))
            self._fwd[slot] = hit
        return hit[1]

    def split3(self, name: str, p: torch.Tensor, transposed: bool) -> torch.Tensor:
        """ME_BF16X3 right-operand copy of a weight ([out, 3 in]; transposed: of W^T, [in, 3 out] for the dgrad GEMMs): the
        fp32-accurate mode's compute copies, rebuilt when the parameter changes (same keys as the other copies)."""
        kind = "x3t" if transposed else "x3"
        k = self._key(p, kind)
        slot = (name, transposed, p.device)
        hit = self._x3.get(slot)
        if hit is not None and hit[0] == k:
            return hit[1]
        if p.dtype != torch.float32 or not p.is_contiguous():
            w32 = ops.cast(p.detach().contiguous(), torch.float32)
            w32 = ops.transpose_cast(w32, torch.float32) if transposed else w32
            out = ops.split3(w32, right_operand=True)
            self._x3[slot] = (k, out)
            return out
        # refresh, in ONE launch, this copy and every stale copy of the same kind that some live Block on the device has used before
        # (after an optimizer step that is all of them: 2 launches per step for the encoder instead of 12 per Block)
        todo = [(self, name, k, p)]
        for c in list(_WeightCache._live):
            for n, w in c._weights().items():
                if (c is self and n == name) or w.device != p.device or w.dtype != torch.float32 or w.dim() != 2 or not w.is_contiguous():
                    continue
                h = c._x3.get((n, transposed, w.device))
                if h is None:
                    continue
                kk = _WeightCache._key(w, kind)
                if h[0] != kk:
                    todo.append((c, n, kk, w))
        outs = ops.split3_many([w.detach() for _, _, _, w in todo], transposed)
        for (c, n, kk, w), t in zip(todo, outs):
            c._x3[(n, transposed, w.device)] = (kk, t)
        return outs[0]

    def folded(self, name: str, w: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor, bias: Optional[torch.Tensor], dtype):
        """LayerNorm folded into the Linear behind it (inference; me_gemm_desc.row_affine):
            Linear(LN(x))[m, n] = rstd_m (x W'^T)[m, n] - rstd_m mean_m s[n] + c[n]
        with W' = gamma o W rounded to the compute dtype, s[n] = sum_k W'[n, k] of the ROUNDED values (so that the mean
        component cancels exactly as the kernel sees it) and c = W beta + bias from the fp32 masters.  One-off weight
        preparation in torch (plumbing), cached like the other compute copies."""
        k = tuple(self._key(t, dtype) for t in (w, ln_w, ln_b) + ((bias,) if bias is not None else ()))
        slot = (name, w.device)
        hit = self._fold.get(slot)
        if hit is None or hit[0] != k:
            with torch.no_grad():
                # every number here comes out of the library (this used to be at::native mul / reduce kernels):
                #   W' = gamma o W rounded to the compute dtype   -- the column-scale form of me_dropout_add (p = 0)
                #   s  = W' 1   (of the ROUNDED values, fp32 sums) -- me_gemm: four rows of ones x W'^T, row 0
                #   c  = W beta + bias                             -- exact-fp32 me_gemm: four copies of beta x W^T + the bias epilogue
                w32 = ops.cast(w.detach().contiguous(), torch.float32)
                g32, b32 = ops._f32(ln_w.detach()).contiguous(), ops._f32(ln_b.detach()).contiguous()
                wf = ops.dropout_add(w32, None, w32.shape[0], 0.0, 0.0, 0, out_dtype=dtype, colscale=g32)
                K = w32.shape[1]
                ones = torch.ones((4, K), dtype=dtype, device=w.device)
                s_vec = ops.gemm(ones, wf, out_dtype=torch.float32)[0].contiguous()
                c_vec = ops.gemm(b32[None, :].expand(4, K).contiguous(), w32, bias=None if bias is None else ops._f32(bias.detach()),
                                 out_dtype=torch.float32)[0].contiguous()
                hit = (k, (wf, s_vec, c_vec))
            self._fold[slot] = hit
        return hit[1]

    _pending = {}         # device -> event behind a refresh of the transposed copies that ran on another stream (prefetch_transposed)

    @classmethod
    def prefetch_transposed(cls, device) -> int:
        """Rebuild, on the CURRENT stream, every transposed copy that exists on `device` and is stale (one batched launch per
        dtype) -- FusedAdamW(overlap=True) calls this on its side stream right behind an optimizer step, so that the 0.15 ms of weight
        transposes run under the next forward instead of at the head of the next backward.  The first `transposed()` hit afterwards
        makes its stream wait for the event recorded here.  Returns the number of copies rebuilt."""
        if cls._live is None:
            return 0
        by_dtype = {}
        for c in list(cls._live):
            for n, w in c._weights().items():
                if w.device != device or w.dim() != 2:
                    continue
                h = c._tr.get((n, w.device))
                if h is None:
                    continue                        # never used in a backward: nothing to keep fresh
                dtype = h[1].dtype
                kk = cls._key(w, dtype)
                if h[0] != kk:
                    by_dtype.setdefault((w.dtype, dtype), []).append((c, n, kk, w))
        n_done = 0
        for (_, dtype), todo in by_dtype.items():
            outs = ops.transpose_cast_many([w.detach().contiguous() for _, _, _, w in todo], dtype)
            for (c, n, kk, w), t in zip(todo, outs):
                c._tr[(n, w.device)] = (kk, t)
            n_done += len(todo)
        if n_done:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
            cls._pending[device] = ev
        return n_done

    def _weights(self):
        blk = self._owner() if self._owner is not None else None
        if blk is None:
            return {}
        return {"qkv": blk.attn.qkv.weight, "proj": blk.attn.proj.weight, "fc1": blk.mlp.fc1.weight, "fc2": blk.mlp.fc2.weight}

    def transposed(self, name: str, p: torch.Tensor, dtype) -> torch.Tensor:
        k = self._key(p, dtype)
        slot = (name, p.device)
        ev = _WeightCache._pending.pop(p.device, None)
        if ev is not None:                      # copies rebuilt on the optimizer's side stream (prefetch_transposed): order this stream behind it
            torch.cuda.current_stream(p.device).wait_event(ev)
        hit = self._tr.get(slot)
        if hit is not None and hit[0] == k:
            return hit[1]
        # refresh every stale transposed copy on this device in one go
        todo = []
        for c in list(_WeightCache._live):
            for n, w in c._weights().items():
                if w.device == p.device and w.dtype == p.dtype and w.dim() == 2:
                    kk = _WeightCache._key(w, dtype)
                    h = c._tr.get((n, w.device))
                    if h is None or h[0] != kk:
                        todo.append((c, n, kk, w))
        if not any(c is self and n == name and kk == k for c, n, kk, _ in todo):
            todo.append((self, name, k, p))       # (a DataParallel replica: its parameters are not the owner's)
        outs = ops.transpose_cast_many([w.detach().contiguous() for _, _, _, w in todo], dtype)
        mine = None
        for (c, n, kk, w), t in zip(todo, outs):
            c._tr[(n, w.device)] = (kk, t)
            if c is self and n == name and kk == k:
                mine = t        # returned from this local: the shared dict may be rewritten by another replica's thread
        return mine


_exact_notes = set()


def _note_exact_fp32(blk, cdt, rdt, reason: str) -> None:
    """fp32_mode = "3xbf16" (the default) covers the plain one-call Block with widths that are multiples of 256; every other fp32 Block runs
    the exact-fp32 kernels -- correct, ~2.5x slower.  Say so ONCE per reason instead of silently (missing #6)."""
    if cdt != torch.float32 or rdt != torch.float32 or reason in _exact_notes:
        return
    if not (blk.fp32_mode == "3xbf16" or blk.compute_dtype == "fp32_3xbf16"):
        return
    _exact_notes.add(reason)
    warnings.warn(f"metatransformer_amd: an fp32 Block runs on the exact-fp32 kernels although fp32_mode = '3xbf16': {reason} "
                  "(the three-product mode covers plain blocks -- no window, no stochastic path, no layer-scale gradient -- with dim and hidden "
                  "multiples of 256)", stacklevel=3)


class _BlockFn(torch.autograd.Function):
    """forward + backward of one Block on the HIP kernels.  Restates Block.forward (attention.py:55-58)."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b, g1, g2, blk, cdt, stoch, grad_mode,
                win=None):
        B, N, C = x.shape
        H = blk.attn.num_heads
        hd = C // H
        M = B * N
        in_dtype = x.dtype
        if in_dtype == torch.float16:      # fp16 exists at the boundary only: residual stream runs in bf16 (same 16 bits,
            x = ops.cast(x, torch.bfloat16)    # fp32's range -- no loss scaling needed), converted back on the way out
        rdt = x.dtype                      # residual-stream dtype
        cache: _WeightCache = blk._wcache
        # grad mode is always off inside Function.forward and needs_input_grad ignores torch.no_grad(): the caller
        # samples torch.is_grad_enabled() and passes it in, so inference saves nothing (no stats / LSE / pre-activation)
        need_grad = grad_mode and any(ctx.needs_input_grad)
        x2 = x.reshape(M, C)
        ctx.fast = False
        fp8 = bool(getattr(blk, "attn_fp8", False)) and win is None and cdt == torch.bfloat16 and hd == 64 and stoch is None
        if stoch is None and win is None and blk.c_side and not fp8 and not (need_grad and g1 is not None):
            # plain path: the whole block is ONE library call (me_block_fwd), the launch sequence lives on the C side
            x3 = blk.uses_3xbf16(cdt, rdt)
            if not x3:
                _note_exact_fp32(blk, cdt, rdt, "dim / hidden not multiples of 256")
            d, keep = _BlockFn._desc(blk, cache, cdt, rdt, B, N, C, H, (n1w, n1b, n2w, n2b, qkvw, qkvb, projw, projb,
                                                                        fc1w, fc1b, fc2w, fc2b, g1, g2), False,
                                     fold=not need_grad and blk.fold_norm, x3=x3)
            # LayerNorm statistics handed from block to block (folded inference): the caller (Block.forward) passes the pairs the
            # previous block left on this very tensor, and gets this block's own back through blk._stats_out
            xs = getattr(blk, "_stats_in", None)
            blk._stats_in = None
            want = keep[3] is not None and blk.chain_stats       # (the folded route is taken)
            y, saved, ys = ops.block_fwd(d, x2, keep=need_grad, x_stats=xs if want else None, want_stats=want)
            blk._stats_out = ys if in_dtype == rdt else None      # (an fp16 caller gets a converted copy: other values)
            del keep
            if need_grad:
                ctx.save_for_backward(x2, saved, n1w, n1b, n2w, n2b, qkvw, qkvb, projw, projb, fc1w, fc1b, fc2w, fc2b)
                ctx.blk, ctx.cdt, ctx.dims, ctx.in_dtype, ctx.fast, ctx.x3 = blk, cdt, (B, N, C, H, hd), in_dtype, True, x3
            return ops.cast(y, in_dtype).reshape(B, N, C)

        _note_exact_fp32(blk, cdt, rdt, "windowed attention" if win is not None else "stochastic path (drop-path / dropout)" if stoch is not None
                         else "layer-scale gradient" if (need_grad and g1 is not None) else "op-by-op composition (c_side = False)")
        xn1, mean1, rstd1 = ops.layernorm_fwd(x2, n1w, n1b, blk.eps, cdt, save_stats=need_grad)
        qkv = ops.gemm(xn1, cache.fwd("qkv", qkvw, cdt), bias=qkvb)
        p_attn = stoch[3] if stoch is not None else 0.0      # training-mode attn_drop (attention.py:33)
        if win is None:
            o, lse = ops.attention_fwd(qkv, B, N, H, hd, blk.attn.scale, need_lse=need_grad, p_drop=p_attn, fp8=fp8,
                                       seed=stoch[2] + 4 if stoch is not None else 0)
            o_att = o
        else:
            # windowed attention (Image/detection/.../base/vit.py:160-190): qkv rows regrouped into ws x ws windows
            # (zero rows where the padded grid exceeds the image), plain attention per window, rows gathered back
            gh_, gw_, ws = win
            nwin = -(-gh_ // ws) * -(-gw_ // ws)
            qkv = ops.window_rows(qkv, B, gh_, gw_, ws, merge=False)
            o_att, lse = ops.attention_fwd(qkv, B * nwin, ws * ws, H, hd, blk.attn.scale, need_lse=need_grad, p_drop=p_attn,
                                           seed=stoch[2] + 4 if stoch is not None else 0)
            o = ops.window_rows(o_att, B, gh_, gw_, ws, merge=True)
        # layer-scale with gradients: d gamma = colsum(dy * UNSCALED branch output), so the branch output is kept and
        # gamma is applied by the residual kernel instead of the GEMM epilogue
        ls_grad = need_grad and g1 is not None
        p_drop, p_path, seed = stoch[:3] if stoch is not None else (0.0, 0.0, 0)
        t1 = t2 = None
        if stoch is None and not ls_grad:
            x1 = ops.gemm(o, cache.fwd("proj", projw, cdt), bias=projb, residual=x2, out_dtype=rdt, colscale=g1)
        else:                 # training-mode proj_drop / drop_path: x1 = x + drop_path(gamma1 * dropout(proj(o)))
            t1 = ops.gemm(o, cache.fwd("proj", projw, cdt), bias=projb, out_dtype=rdt, colscale=None if ls_grad else g1)
            x1 = ops.dropout_add(t1, x2, N, p_drop, p_path, seed + 1, colscale=g1 if ls_grad else None)
        xn2, mean2, rstd2 = ops.layernorm_fwd(x1, n2w, n2b, blk.eps, cdt, save_stats=need_grad)
        hpre = torch.empty((M, fc1w.shape[0]), dtype=cdt, device=x.device) if need_grad else None
        a = ops.gemm(xn2, cache.fwd("fc1", fc1w, cdt), bias=fc1b, act=ME_ACT_GELU, preact=hpre,
                     flags=ME_GEMM_SAVE_GELU_GRAD if need_grad else 0)      # saved: gelu'(h) (as me_block_fwd does)
        if stoch is None and not ls_grad:
            y = ops.gemm(a, cache.fwd("fc2", fc2w, cdt), bias=fc2b, residual=x1, out_dtype=rdt, colscale=g2)
        else:                 # Mlp: fc1 -> act -> drop -> fc2 -> drop (mlp.py:29-35), then drop_path + residual
            if p_drop > 0:
                a = ops.dropout_add(a, None, N, p_drop, 0.0, seed + 2)
            t2 = ops.gemm(a, cache.fwd("fc2", fc2w, cdt), bias=fc2b, out_dtype=rdt, colscale=None if ls_grad else g2)
            y = ops.dropout_add(t2, x1, N, p_drop, p_path, seed + 3, colscale=g2 if ls_grad else None)
        if not ls_grad:
            t1 = t2 = None

        if need_grad:
            ctx.save_for_backward(x2, mean1, rstd1, xn1, qkv, lse, o, x1, mean2, rstd2, xn2, hpre, a,
                                  n1w, qkvw, projw, n2w, fc1w, fc2w, g1, g2, t1, t2, o_att if win is not None else None)
            ctx.blk, ctx.cdt, ctx.dims, ctx.stoch, ctx.win = blk, cdt, (B, N, C, H, hd), stoch, win
            ctx.has_bias = (qkvb is not None, projb is not None, fc1b is not None, fc2b is not None)
            ctx.in_dtype = in_dtype
        return ops.cast(y, in_dtype).reshape(B, N, C)

    @staticmethod
    def _desc(blk, cache, cdt, rdt, B, N, C, H, params, transposed, fold=False, x3=False):
        """me_block_desc for this block + the list of tensors that must stay alive while the call is in flight.
        fold: inference with both LayerNorms folded into qkv / fc1 (bf16 compute on a bf16 token stream; the library falls
        back to LayerNorm + GEMM by itself when the descriptor carries no folded weights)."""
        (n1w, n1b, n2w, n2b, qkvw, qkvb, projw, projb, fc1w, fc1b, fc2w, fc2b, g1, g2) = params
        ws = {"qkv": qkvw, "proj": projw, "fc1": fc1w, "fc2": fc2w}
        if x3:            # fp32-accurate mode: three-plane bf16 copies of the fp32 weights (Block.fp32_mode = "3xbf16")
            w = {k: cache.split3(k, t, False) for k, t in ws.items()}
            wt = {k: cache.split3(k, t, True) for k, t in ws.items()} if transposed else None
        else:
            w = {k: cache.fwd(k, t, cdt) for k, t in ws.items()}
            wt = {k: cache.transposed(k, t, cdt) for k, t in ws.items()} if transposed else None
        f = lambda t: None if t is None else ops._f32(t).contiguous()      # noqa: E731
        vec = dict(ln1_g=f(n1w), ln1_b=f(n1b), ln2_g=f(n2w), ln2_b=f(n2b), qkv_b=f(qkvb), proj_b=f(projb), fc1_b=f(fc1b),
                   fc2_b=f(fc2b), gamma1=f(g1), gamma2=f(g2))
        d = ops.block_desc(B, N, C, H, fc1w.shape[0], blk.eps, blk.attn.scale, cdt, rdt, w, wt, vec, x3=x3)
        folded = None
        # (only when every CU gets 256 x 256 tiles of qkv: the fold's row-affine epilogue lives in the resident GEMM kernel; below
        #  that the generic epilogue it would fall back to measured slower than LayerNorm + GEMM -- B = 32: 2.95 vs 2.75 ms)
        big = ((B * N + 255) // 256) * ((3 * C + 255) // 256) >= 256
        if fold and (big or fold == "always") and cdt == torch.bfloat16 and rdt == torch.bfloat16:
            folded = (cache.folded("qkv", qkvw, n1w, n1b, qkvb, cdt), cache.folded("fc1", fc1w, n2w, n2b, fc1b, cdt))
            (d.qkv_wf, d.qkv_s, d.qkv_c), (d.fc1_wf, d.fc1_s, d.fc1_c) = [tuple(ops.ptr(t) for t in trip) for trip in folded]
        return d, (w, wt, vec, folded)

    @staticmethod
    def _backward_c(ctx, dy):
        (x2, saved, n1w, n1b, n2w, n2b, qkvw, qkvb, projw, projb, fc1w, fc1b, fc2w, fc2b) = ctx.saved_tensors
        blk, cdt = ctx.blk, ctx.cdt
        B, N, C, H, hd = ctx.dims
        rdt = x2.dtype
        ng = ctx.needs_input_grad
        dy2 = dy.contiguous().reshape(B * N, C)
        if dy2.dtype != rdt:
            dy2 = ops.cast(dy2, rdt)
        d, keep = _BlockFn._desc(blk, blk._wcache, cdt, rdt, B, N, C, H, (n1w, n1b, n2w, n2b, qkvw, qkvb, projw, projb,
                                                                          fc1w, fc1b, fc2w, fc2b, None, None), True, x3=ctx.x3)
        # gradient destinations: (name in me_block_grads, parameter, index into needs_input_grad)
        slots = [("ln1_g", blk.norm1.weight, 1), ("ln1_b", blk.norm1.bias, 2), ("qkv_w", blk.attn.qkv.weight, 3),
                 ("qkv_b", blk.attn.qkv.bias, 4), ("proj_w", blk.attn.proj.weight, 5), ("proj_b", blk.attn.proj.bias, 6),
                 ("ln2_g", blk.norm2.weight, 7), ("ln2_b", blk.norm2.bias, 8), ("fc1_w", blk.mlp.fc1.weight, 9),
                 ("fc1_b", blk.mlp.fc1.bias, 10), ("fc2_w", blk.mlp.fc2.weight, 11), ("fc2_b", blk.mlp.fc2.bias, 12)]
        slots = [(n, p, i) for n, p, i in slots if p is not None and ng[i]]
        # in place into a FlatParams buffer (accumulating) when EVERY wanted gradient lives there, else fresh tensors
        flat = getattr(blk.attn.qkv.weight, "_me_flat", None)
        direct = None
        if flat is not None and slots and all(p.dtype == torch.float32 and getattr(p, "_me_flat", None) is flat for _, p, _ in slots):
            direct = [flat.direct_grad(p) for _, p, _ in slots]
            if any(t is None for t in direct):
                direct = None
        g = _capi.BlockGrads()
        wdt = blk.attn.qkv.weight.dtype
        g.w_dtype = _capi.dtype_code(wdt if wdt != torch.float16 else torch.float32)
        g.accumulate = 1 if direct is not None else 0
        outs = {}
        for j, (n, p, i) in enumerate(slots):
            if direct is not None:
                t = direct[j]
            elif n.endswith("_w"):
                t = torch.empty(p.shape, dtype=wdt if wdt != torch.float16 else torch.float32, device=p.device)
            else:
                t = torch.empty(p.shape, dtype=torch.float32, device=p.device)
            outs[i] = t
            setattr(g, n, ops.ptr(t))
        dx = ops.block_bwd(d, x2, dy2, saved, g)
        del keep
        res = [None] * 20
        res[0] = ops.cast(dx, ctx.in_dtype).reshape(B, N, C) if ng[0] else None
        for n, p, i in slots:
            if direct is not None:
                flat.grad_written(p)
            else:
                t = outs[i]
                res[i] = t if t.dtype == p.dtype else (ops.cast(t, p.dtype) if t.dim() == 2 else t.to(p.dtype))
        return tuple(res)

    @staticmethod
    def backward(ctx, dy):
        if ctx.fast:
            return _BlockFn._backward_c(ctx, dy)
        (x2, mean1, rstd1, xn1, qkv, lse, o, x1, mean2, rstd2, xn2, hpre, a,
         n1w, qkvw, projw, n2w, fc1w, fc2w, g1, g2, t1, t2, o_att) = ctx.saved_tensors
        blk, cdt = ctx.blk, ctx.cdt
        B, N, C, H, hd = ctx.dims
        M = B * N
        cache: _WeightCache = blk._wcache
        rdt = x2.dtype
        ng = ctx.needs_input_grad      # x, n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b, g1, g2
        dy2 = dy.contiguous().reshape(M, C)
        if dy2.dtype != rdt:
            dy2 = ops.cast(dy2, rdt)

        def wgrad(dout, inp, lin, need_w, need_b):
            """dW[out,in] = dout^T inp in the parameter's dtype, db = column sums of dout (fused into the same kernel).
            When the parameters live in a parallel.FlatParams the weight gradient is ACCUMULATED in place into its flat
            view by the GEMM epilogue (beta = 1) and autograd gets None for it -- no separate `grad += dW` pass."""
            w = lin.weight
            if not need_w:
                return None, (ops.colsum(dout).to(w.dtype) if need_b else None)
            flat = getattr(w, "_me_flat", None)
            tgt = flat.direct_grad(w) if (flat is not None and w.dtype == torch.float32) else None
            tgt_b = None
            if tgt is not None and need_b and getattr(lin.bias, "_me_flat", None) is flat and lin.bias.dtype == torch.float32:
                tgt_b = flat.direct_grad(lin.bias)
            # the fused column sums share C's beta: usable when both accumulate in place or neither does
            fuse_b = need_b and (tgt is None or tgt_b is not None)
            odt = w.dtype if w.dtype != torch.float16 else torch.float32      # fp16 parameters: fp32 result, cast below
            res = ops.gemm(dout, inp, op=ME_GEMM_TN, out=tgt, out_dtype=odt, beta=1.0 if tgt is not None else 0.0,
                           want_colsum_a=fuse_b, colsum_out=tgt_b)
            dw, db = res if fuse_b else (res, ops.colsum(dout) if need_b else None)
            if tgt is not None:
                flat.grad_written(w)
                dw = None
            if tgt_b is not None:
                flat.grad_written(lin.bias)
                db = None
            if dw is not None and dw.dtype != w.dtype:
                dw = ops.cast(dw, w.dtype)
            return dw, (db.to(w.dtype) if db is not None else None)

        def ln_bwd(dyn, xin, mean, rstd, norm, dres, need_aff):
            """LayerNorm backward; affine gradients accumulated in place when the parameters live in a FlatParams"""
            flat = getattr(norm.weight, "_me_flat", None)
            acc = None
            if need_aff and flat is not None and getattr(norm.bias, "_me_flat", None) is flat and norm.weight.dtype == torch.float32:
                gw, gb = flat.direct_grad(norm.weight), flat.direct_grad(norm.bias)
                if gw is not None and gb is not None:
                    acc = (gw, gb)
            dxo, dg, db = ops.layernorm_bwd(dyn, xin, mean, rstd, norm.weight, dres, rdt, need_aff, affine_accum=acc)
            if acc is not None:
                flat.grad_written(norm.weight)
                flat.grad_written(norm.bias)
            return dxo, dg, db

        stoch = ctx.stoch
        p_drop, p_path, seed = stoch[:3] if stoch is not None else (0.0, 0.0, 0)
        p_attn = stoch[3] if stoch is not None else 0.0

        def branch_grad(dout, t, g, sd):
            """gradient entering a residual branch y = x + drop_path(gamma * dropout(t)): (d t in compute dtype, d gamma).
            Stochastic training: the SAME masks, regenerated from the seed; layer-scale: d gamma = colsum(masked dy * t)."""
            dm = dout if stoch is None else ops.dropout_add(dout, None, N, p_drop, p_path, sd)
            if g is None:
                return ops.cast(dm, cdt), None
            dg = ops.colsum_mul(dm, t)
            return ops.dropout_add(dm, None, N, 0.0, 0.0, 0, out_dtype=cdt, colscale=g), dg

        # ---- MLP branch: y = x1 + gamma2 * fc2(gelu(fc1(LN2(x1))))
        dy_c, d_g2 = branch_grad(dy2, t2, g2, seed + 3)
        dh = ops.gemm(dy_c, cache.transposed("fc2", fc2w, cdt), aux=hpre, flags=ME_GEMM_AUX_IS_FACTOR)   # dA * gelu'(h)
        if stoch is not None and p_drop > 0:
            dh = ops.dropout_add(dh, None, N, p_drop, 0.0, seed + 2)
        d_fc2w, d_fc2b = wgrad(dy_c, a, blk.mlp.fc2, ng[11], ng[12] and ctx.has_bias[3])
        dxn2 = ops.gemm(dh, cache.transposed("fc1", fc1w, cdt))
        d_fc1w, d_fc1b = wgrad(dh, xn2, blk.mlp.fc1, ng[9], ng[10] and ctx.has_bias[2])
        dx1, d_n2w, d_n2b = ln_bwd(dxn2, x1, mean2, rstd2, blk.norm2, dy2, ng[7] or ng[8])

        # ---- attention branch: x1 = x + proj(attn(qkv(LN1(x))))
        dx1_c, d_g1 = branch_grad(dx1, t1, g1, seed + 1)
        do = ops.gemm(dx1_c, cache.transposed("proj", projw, cdt))
        d_projw, d_projb = wgrad(dx1_c, o, blk.attn.proj, ng[5], ng[6] and ctx.has_bias[1])
        if ctx.win is None:
            dqkv = ops.attention_bwd(qkv, o, do, lse, B, N, H, hd, blk.attn.scale, p_drop=p_attn, seed=seed + 4)
        else:                 # the same regrouping on the gradient; padded rows are constants (no gradient leaves them)
            gh_, gw_, ws = ctx.win
            nwin = -(-gh_ // ws) * -(-gw_ // ws)
            do_w = ops.window_rows(do, B, gh_, gw_, ws, merge=False)
            dqkv_w = ops.attention_bwd(qkv, o_att, do_w, lse, B * nwin, ws * ws, H, hd, blk.attn.scale, p_drop=p_attn,
                                       seed=seed + 4)
            dqkv = ops.window_rows(dqkv_w, B, gh_, gw_, ws, merge=True)
        dxn1 = ops.gemm(dqkv, cache.transposed("qkv", qkvw, cdt))
        d_qkvw, d_qkvb = wgrad(dqkv, xn1, blk.attn.qkv, ng[3], ng[4] and ctx.has_bias[0])
        dx, d_n1w, d_n1b = ln_bwd(dxn1, x2, mean1, rstd1, blk.norm1, dx1, ng[1] or ng[2])

        def aff(g, p):
            return None if g is None else g.to(p.dtype)

        return (ops.cast(dx, ctx.in_dtype).reshape(B, N, C) if ng[0] else None,
                aff(d_n1w, n1w) if ng[1] else None, aff(d_n1b, n1w) if ng[2] else None,
                d_qkvw, d_qkvb, d_projw, d_projb,
                aff(d_n2w, n2w) if ng[7] else None, aff(d_n2b, n2w) if ng[8] else None,
                d_fc1w, d_fc1b, d_fc2w, d_fc2b,
                d_g1.to(g1.dtype) if (d_g1 is not None and ng[13]) else None,
                d_g2.to(g2.dtype) if (d_g2 is not None and ng[14]) else None, None, None, None, None, None)


class Block(nn.Module):
    """timm.models.vision_transformer.Block, served by HIP kernels.

    Signature follows timm 0.4.12 (the version the reference pins); ``layer_scale`` adds the per-channel
    gamma1/gamma2 of the Image pipelines (Image/detection/mmdet_custom/models/backbones/base/vit.py:298-320);
    ``windowed`` / ``window_size`` select that file's WindowedAttention (:148-192), whose forward takes the token grid:
    ``blk(x, H, W)`` (the detection backbone calls every block that way, :312-329; H, W are ignored by global blocks).
    """

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, layer_scale=False, windowed=False,
                 window_size=14):
        super().__init__()
        self.windowed, self.window_size = bool(windowed), int(window_size)
        self.eps = _resolve_eps(norm_layer)
        self.norm1 = nn.LayerNorm(dim, eps=self.eps)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                              attn_drop=attn_drop, proj_drop=drop)
        self.drop_path_prob = float(drop_path)
        self.drop_path = nn.Identity()          # name kept for state_dict / repr parity (no parameters)
        self.norm2 = nn.LayerNorm(dim, eps=self.eps)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.layer_scale = bool(layer_scale)
        if layer_scale:
            self.gamma1 = nn.Parameter(torch.ones(dim))
            self.gamma2 = nn.Parameter(torch.ones(dim))
        self.compute_dtype: Optional[torch.dtype] = None     # override; None = infer (autocast / param dtype)
        self.c_side = True          # plain blocks run as one me_block_fwd / me_block_bwd call; False = op-by-op composition
        self.attn_fp8 = False       # True: e4m3 attention forward (me_attention_fwd_fp8; bf16 compute, head_dim 64) -- config 5
        self.fold_norm = True       # (True: when it pays, see _desc; "always": whenever it is legal; False: never)  inference (no gradient wanted), bf16 compute on a bf16 token stream: norm1 / norm2 folded into
                                    # qkv / fc1 (me_row_stats + row_affine GEMM epilogue instead of me_layernorm_fwd + GEMM)
        # fp32 compute: "3xbf16" (the default since round 6: Block.default_fp32_mode) = fp32-accurate arithmetic on the bf16 matrix pipe --
        # three bf16 products per Linear and per attention product on hi / lo split operands, ~1e-5 of the reference (bound 1e-4; the
        # north star's fp32 bound is 1e-3), 2.3 .. 2.9x the exact path on every fp32 recipe of the reference (all of which run the plain
        # Block: frozen encoder, no dropout -- SURVEY appendix B); "exact" = the exact-fp32 MFMA (157 TF peak; ~1e-6).  The three-product
        # form covers plain blocks with C and hidden multiples of 256; windowed / stochastic / layer-scale-gradient paths and other
        # widths run exact whatever the mode (uses_3xbf16 tells).  Also selected by compute_dtype = "fp32_3xbf16".
        self.fp32_mode = Block.default_fp32_mode
        self._wcache = _WeightCache()
        self.chain_stats = True     # folded inference: LayerNorm statistics from the proj / fc2 epilogues, handed from block to block
        self._stats_in = self._stats_out = None      # (per-call hand-over between forward() and the autograd Function)

    default_fp32_mode = "3xbf16"      # what a new Block's fp32_mode starts as; `Block.default_fp32_mode = "exact"` (or set_fp32_mode(model, "exact")) restores round 5's

    def uses_3xbf16(self, cdt, rdt) -> bool:
        """does the plain (one-call) path of this block run in the fp32-accurate three-product mode for these dtypes?"""
        want = self.fp32_mode == "3xbf16" or self.compute_dtype == "fp32_3xbf16"
        if self.fp32_mode not in ("exact", "3xbf16"):
            raise MetaEncError(f"Block.fp32_mode must be 'exact' or '3xbf16' (got {self.fp32_mode!r})")
        C, hidden = self.attn.qkv.weight.shape[1], self.mlp.fc1.weight.shape[0]
        return bool(want and cdt == torch.float32 and rdt == torch.float32 and C % 256 == 0 and hidden % 256 == 0)

    def _compute_dtype(self, x: torch.Tensor) -> torch.dtype:
        if self.compute_dtype == "fp32_3xbf16":
            return torch.float32
        if self.compute_dtype is not None:
            return self.compute_dtype
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")
            if dt not in (torch.bfloat16, torch.float16):
                raise MetaEncError(f"autocast dtype {dt} unsupported: libmetaenc computes in bfloat16 or float32")
            return torch.bfloat16           # fp16 autocast (Audio/src/traintest.py, mmcv fp16) runs on the bf16 kernels
        dt = self.attn.qkv.weight.dtype
        return torch.bfloat16 if dt == torch.float16 else dt      # .half() checkpoints: bf16 compute, fp16 in / out

    def forward(self, x: torch.Tensor, H: Optional[int] = None, W: Optional[int] = None) -> torch.Tensor:
        win = None
        if self.windowed:
            if H is None or W is None or H * W != x.shape[1]:
                raise MetaEncError(f"windowed Block needs the token grid: blk(x, H, W) with H*W == N (got H={H}, W={W}, "
                                   f"N={x.shape[1] if x.dim() == 3 else '?'})")
            win = (int(H), int(W), self.window_size)
        if x.dim() != 3:
            raise MetaEncError(f"Block expects [B, N, C] tokens, got shape {tuple(x.shape)}")
        if not x.is_cuda:
            raise MetaEncError("metatransformer_amd.Block runs on MI355X only: input is a CPU tensor and there is "
                               "no CPU fallback (use oracle/ for CPU reference numbers)")
        stoch = None
        if self.training:
            if self.attn.proj_drop.p != self.mlp.drop.p:
                raise MetaEncError("proj_drop and mlp drop must be equal (timm's Block passes one `drop` to both)")
            if self.mlp.drop.p > 0 or self.drop_path_prob > 0 or self.attn.attn_drop.p > 0:
                # one seed per call from torch's CPU generator (reproducible under torch.manual_seed)
                seed = int(torch.empty((), dtype=torch.int64).random_().item())
                stoch = (float(self.mlp.drop.p), self.drop_path_prob, seed, float(self.attn.attn_drop.p))
        self._wcache.bind(self)
        cdt = self._compute_dtype(x)
        if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise MetaEncError(f"unsupported token dtype {x.dtype}")
        if x.shape[0] == 0 or x.shape[1] == 0:
            # an empty batch (the last, ragged step of a data loader with drop_last = False on another rank) or no tokens: the reference's Block
            # returns the empty tensor; so does this one, without a launch (the parameters receive no gradient from it)
            return x.clone()
        x = x.contiguous()
        a, m = self.attn, self.mlp
        g1 = self.gamma1 if self.layer_scale else None
        g2 = self.gamma2 if self.layer_scale else None
        # folded inference: the LayerNorm statistics of a block's output come out of its fc2 epilogue and ride on the output
        # TENSOR OBJECT (attribute _me_ln_stats = (pairs, eps, version)) to whichever Block is handed that same object next --
        # nn.Sequential, a `for blk in blocks` loop.  Anything else (a new tensor from x + pos, an in-place edit: the version
        # moves) simply finds no statistics and reads its input once more (me_row_stats).
        # Inference tensors (torch.inference_mode) have no version counter: nothing can vouch for "not written since", so they
        # are neither tagged nor trusted (the block reads its input once more; encoder_forward_inference chains on the C side).
        # Assumptions of the hand-over: (1) whoever writes the tagged tensor bumps its `_version` -- torch ops do; a write
        # through a raw pointer (this library's own `out=` forms, `x.data` edits, foreign kernels) does not, and must drop the tag
        # (`del y._me_ln_stats`) or set `blk.chain_stats = False`; (2) one forward at a time per Block object: `_stats_in / _stats_out`
        # are per-call state on the module -- a Block shared by two host threads needs `chain_stats = False` (me_encoder_fwd, the
        # one-call route, keeps the hand-over inside the library and has neither restriction).
        tag = getattr(x, "_me_ln_stats", None)
        self._stats_in = None
        ver = _tensor_version(x) if tag is not None else None
        if (tag is not None and ver is not None and self.chain_stats and tag[1] == self.eps and tag[2] == ver
                and tag[0].shape[-2] == x.shape[0] * x.shape[1]):
            self._stats_in = tag[0]
        self._stats_out = None
        y = _BlockFn.apply(x, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight,
                           a.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias,
                           m.fc2.weight, m.fc2.bias, g1, g2, self, cdt, stoch, torch.is_grad_enabled(), win)
        self._stats_in = None
        if self._stats_out is not None:
            ver = _tensor_version(y)
            if ver is not None:
                y._me_ln_stats = (self._stats_out, self.eps, ver)
            self._stats_out = None
        return y


def resize_pos_embed(pos_embed: torch.Tensor, input_shape, pos_shape, mode: str = "bicubic") -> torch.Tensor:
    """TIMMVisionTransformer.resize_pos_embed (Image/detection/mmdet_custom/models/backbones/base/vit.py:459-486), same
    signature: pos_embed [1, L, C] whose LAST pos_h * pos_w rows are the grid table and whose row 0 is the cls entry ->
    [1, 1 + H * W, C] with the grid resampled (align_corners=False) on the GPU."""
    if pos_embed.dim() != 3 or pos_embed.shape[0] != 1:
        raise MetaEncError("shape of pos_embed must be [1, L, C]")
    if not pos_embed.is_cuda:
        raise MetaEncError("resize_pos_embed runs on MI355X only (no CPU fallback)")
    ph, pw = pos_shape
    grid = pos_embed[0, -ph * pw:].contiguous()
    new = ops.resize_rows(grid, (ph, pw), tuple(input_shape), mode)
    return torch.cat([pos_embed[:, :1], new.unsqueeze(0)], dim=1)


def convert_video_state_dict(sd) -> "dict":
    """Key set of the Video pipeline's blocks (Video/models/modeling_finetune.py) -> this package's Block keys:
        attn.q_bias, attn.v_bias (:160-166; K has no bias, :172-178)  ->  attn.qkv.bias = cat(q_bias, 0, v_bias)
        gamma_1, gamma_2 (:245-259)                                   ->  gamma1, gamma2   (Block(layer_scale=True))
    Works on one block's dict or on a whole `blocks.{i}.`-prefixed checkpoint; other keys pass through.  The zero K third
    stays without effect in training too: a bias on K shifts every score of a softmax row by the same amount."""
    out = {}
    for k, v in sd.items():
        if k.endswith("attn.v_bias"):
            continue
        if k.endswith("attn.q_bias"):
            vb = sd[k[:-len("q_bias")] + "v_bias"]
            out[k[:-len("q_bias")] + "qkv.bias"] = torch.cat([v, torch.zeros_like(vb), vb])
        elif k.endswith("gamma_1"):
            out[k[:-len("gamma_1")] + "gamma1"] = v
        elif k.endswith("gamma_2"):
            out[k[:-len("gamma_2")] + "gamma2"] = v
        else:
            out[k] = v
    return out


def to_video_state_dict(sd) -> "dict":
    """Inverse of convert_video_state_dict (also maps gradients keyed like a state dict)."""
    out = {}
    for k, v in sd.items():
        if k.endswith("attn.qkv.bias"):
            C = v.shape[0] // 3
            out[k[:-len("qkv.bias")] + "q_bias"] = v[:C].clone()
            out[k[:-len("qkv.bias")] + "v_bias"] = v[2 * C:].clone()
        elif k.endswith("gamma1"):
            out[k[:-1] + "_1"] = v
        elif k.endswith("gamma2"):
            out[k[:-1] + "_2"] = v
        else:
            out[k] = v
    return out


def build_encoder(depth: int = 12, dim: int = 768, num_heads: int = 12, mlp_ratio: float = 4., qkv_bias: bool = True,
                  norm_layer=nn.LayerNorm, **kw) -> nn.Sequential:
    """The reference's canonical construction (README.md:124-135):
    Base = (12, 768, 12), Large = (24, 1024, 16).  Returns a plain nn.Sequential so that
    ``load_state_dict(ckpt, strict=True)``, slicing and iteration behave exactly as at the reference call sites."""
    return nn.Sequential(*[Block(dim=dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                 norm_layer=norm_layer, act_layer=nn.GELU, **kw) for _ in range(depth)])


def set_fp32_mode(model: nn.Module, mode: str) -> int:
    """Select the fp32 arithmetic of every Block inside `model` (any module tree: the reference's `nn.Sequential` encoder, or a task model
    that owns one): "3xbf16" -- the default: fp32-accurate (~5e-6 of the reference, bound 1e-4; the north star's fp32 bound is 1e-3) on
    the bf16 matrix pipe, 2.3 .. 2.9x per step on the reference's fp32 recipes (profiles/r05_refshapes.txt) -- or "exact" -- the
    exact-fp32 MFMA (~1e-6): `metatransformer_amd.set_fp32_mode(encoder, "exact")` is the one line that restores it.  Returns the number
    of Blocks set."""
    if mode not in ("exact", "3xbf16"):
        raise MetaEncError(f"fp32 mode must be 'exact' or '3xbf16' (got {mode!r})")
    n = 0
    for m in model.modules():
        if isinstance(m, Block):
            m.fp32_mode = mode
            n += 1
    return n


# encoder_forward_inference replays a captured hipGraph for small batches: below ~1 000 token rows the 84 launches of a Base
# forward are latency-bound and the host's launch path is a visible part of it (N = 197: B = 1 0.84 ms eager, 0.63 ms replayed;
# B = 2 0.87 / 0.68; B = 4 0.84 / 0.76; B = 8 0.99 / 0.99; B = 16 1.29 / 1.34 -- profiles/r05_latency.txt).
GRAPH_MAX_ROWS = 1024
GRAPH_MAX_CACHED = 8      # captured graphs kept per encoder (one per (B, N, C, dtype); each pins its static buffers and workspace)
_graphs = None      # weakref.WeakKeyDictionary: encoder -> {(B, N, C, dtype, device): _EncoderGraph}


class _EncoderGraph:
    """One captured forward: static input / output buffers, the graph, and what it was captured against (the weights' identity:
    a graph bakes device pointers in, so an optimizer step, load_state_dict or .to() re-captures)."""

    def __init__(self, wkey, x_static, y_static, graph):
        self.wkey, self.x, self.y, self.graph = wkey, x_static, y_static, graph


def _encoder_weight_key(blocks):
    return (ops.WEIGHT_EPOCH,) + tuple((p.data_ptr(), _tensor_version(p), p.dtype) for b in blocks for p in b.parameters()) + tuple(
        (b.compute_dtype, b.fold_norm, b.eps, b.fp32_mode) for b in blocks)


def _encoder_descs(blocks, x: torch.Tensor, B: int, N: int, C: int):
    descs, keep = [], []
    for b in blocks:
        b._wcache.bind(b)
        cdt = b._compute_dtype(x)
        a, m = b.attn, b.mlp
        g1 = b.gamma1 if b.layer_scale else None
        g2 = b.gamma2 if b.layer_scale else None
        d, k = _BlockFn._desc(b, b._wcache, cdt, x.dtype, B, N, C, a.num_heads,
                              (b.norm1.weight, b.norm1.bias, b.norm2.weight, b.norm2.bias, a.qkv.weight, a.qkv.bias,
                               a.proj.weight, a.proj.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, g1, g2), False,
                              fold=b.fold_norm, x3=b.uses_3xbf16(cdt, x.dtype))
        descs.append(d)
        keep.append(k)
    return descs, keep


@torch.no_grad()
def encoder_forward_inference(encoder: nn.Sequential, x: torch.Tensor, graph: Optional[bool] = None) -> torch.Tensor:
    """``encoder(x)`` for a plain stack of Blocks in eval mode as ONE library call (me_encoder_fwd): what a serving host that
    is not Python would do.  Falls back to nothing: raises if a block is not a plain Block.

    graph: None (default) = replay a cached hipGraph of that call when B * N <= GRAPH_MAX_ROWS (one graph per (B, N, C, dtype),
    re-captured when a weight changes; input copied into / output cloned out of the graph's static buffers); True / False force it."""
    blocks = list(encoder)
    if not blocks or any(not isinstance(b, Block) or b.windowed for b in blocks):
        raise MetaEncError("encoder_forward_inference: a non-empty nn.Sequential of plain (non-windowed) Blocks is required")
    if x.dim() != 3 or not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16):
        raise MetaEncError("encoder_forward_inference: [B, N, C] fp32 / bf16 CUDA tokens required")
    B, N, C = x.shape
    x2 = x.contiguous().reshape(B * N, C)
    if graph is None:
        graph = B * N <= GRAPH_MAX_ROWS
    if graph and torch.cuda.is_current_stream_capturing():
        graph = False                       # (already inside somebody else's capture: just enqueue)
    if not graph:
        descs, keep = _encoder_descs(blocks, x, B, N, C)
        y = ops.encoder_fwd(descs, x2)
        del keep
        return y.reshape(B, N, C)

    global _graphs
    if _graphs is None:
        import weakref
        _graphs = weakref.WeakKeyDictionary()
    per_enc = _graphs.setdefault(encoder, {})
    #  what the blocks RESOLVE to is part of the key: under torch.autocast the same encoder computes in another dtype, and a
    # graph captured outside autocast must not be replayed inside it (or the reverse).  The static x / y buffers of a cached graph are
    # shared by every caller with the same key: replay is serialised on the calling thread's current stream (copy in, replay, clone out
    # are enqueued on it back to back); concurrent callers on DIFFERENT streams must pass graph=False or use their own encoder object.
    cdts = tuple((str(b._compute_dtype(x)), b.uses_3xbf16(b._compute_dtype(x), x.dtype)) for b in blocks)
    gkey = (B, N, C, x.dtype, x.device, cdts)
    wkey = _encoder_weight_key(blocks)
    g = per_enc.get(gkey)
    if g is None or g.wkey != wkey:
        descs, keep = _encoder_descs(blocks, x, B, N, C)
        xs = torch.empty_like(x2)
        xs.copy_(x2)
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):       # warm-up outside the capture: weight copies, per-stream work counters, lazy attributes
            ops.encoder_fwd(descs, xs)
        torch.cuda.current_stream(x.device).wait_stream(side)
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            ys = ops.encoder_fwd(descs, xs)
        g = _EncoderGraph(wkey, xs, ys, cg)
        g.keep = (descs, keep)              # the compute copies of the weights the graph points at
        per_enc.pop(gkey, None)
        while len(per_enc) >= GRAPH_MAX_CACHED:      # oldest shape out (dicts keep insertion order): a graph pins its buffers
            per_enc.pop(next(iter(per_enc)))
        per_enc[gkey] = g
    g.x.copy_(x2)
    g.graph.replay()
    return g.y.clone().reshape(B, N, C)


def encoder_flops_per_sample(N: int, C: int, L: int) -> float:
    """F(N,C,L) = L * (24 N C^2 + 4 N^2 C)  -- the work model of BASELINE.md section 3 (forward)."""
    return float(L) * (24.0 * N * C * C + 4.0 * N * N * C)
