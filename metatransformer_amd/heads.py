"""Task heads behind the encoder and encoder checkpoint I/O.

Heads restated from the reference pipelines, with the reference parameter names so their checkpoints load strict=True:
  * ``ClassifierHead``  -- Video/models/modeling_finetune.py:445-454 + :395: ``fc_norm(x.mean(1))`` (use_mean_pooling) or
    ``norm(x)[:, 0]``, dropout, ``head`` Linear;
  * ``ClsHead``         -- PointCloud/openpoints/models/classification/cls_base.py:77-136: optional 'max' / 'avg' global
    features over the token axis, then Linear(+BatchNorm1d+ReLU)(+Dropout) blocks and a final Linear.
Token pooling, LayerNorm and every Linear (forward, dgrad, wgrad, bias gradient) run in libmetaenc.so; BatchNorm1d / ReLU /
Dropout on the [B, 256] head activations are PyTorch glue, as the north star prescribes for the heads.

Checkpoint I/O: ``load_encoder_checkpoint`` / ``save_encoder_checkpoint`` move between the reference's wire format (a bare
OrderedDict of ``{i}.norm1.weight ...`` fp32 tensors, README.md:125-135, optionally prefixed ``blocks.`` / ``module.`` /
``encoder.``, or the Video key set) and the device-resident packed layout (``pack_encoder``: one flat fp32 master buffer
whose slices ARE the parameters, its bf16 mirror for the forward GEMMs and the pre-transposed bf16 copies for dgrad).
"""
from __future__ import annotations

import re
from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn

from . import _capi, ops
from ._capi import MetaEncError, check, dtype_code, ptr, stream_ptr
from .encoder import Block, convert_video_state_dict

_POOL = {"mean": _capi.ME_POOL_MEAN, "avg": _capi.ME_POOL_MEAN, "max": _capi.ME_POOL_MAX, "cls": _capi.ME_POOL_FIRST,
         "first": _capi.ME_POOL_FIRST}


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        if x.dim() != 3 or not x.is_cuda:
            raise MetaEncError("pool_tokens: [B, N, C] CUDA tokens required (no CPU fallback)")
        x = x.contiguous()
        B, N, C = x.shape
        if x.dtype == torch.float16:
            x = ops.cast(x, torch.bfloat16)
        out = torch.empty(B, C, dtype=torch.float32, device=x.device)
        need = x.requires_grad and mode == _capi.ME_POOL_MAX
        arg = torch.empty(B, C, dtype=torch.int32, device=x.device) if (need or mode == _capi.ME_POOL_MAX) else None
        check(_capi.load().me_pool_tokens(ptr(x), dtype_code(x.dtype), ptr(out), ptr(arg), B, N, C, mode, stream_ptr()),
              "me_pool_tokens")
        ctx.mode, ctx.shape, ctx.dtype = mode, (B, N, C), x.dtype
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        B, N, C = ctx.shape
        dx = torch.empty(B, N, C, dtype=ctx.dtype, device=dy.device)
        dyc = dy.float().contiguous()
        check(_capi.load().me_pool_tokens_bwd(ptr(dyc), ptr(arg), ptr(dx), dtype_code(dx.dtype), B, N, C, ctx.mode, stream_ptr()),
              "me_pool_tokens_bwd")
        return dx, None


def pool_tokens(x: torch.Tensor, mode: str = "mean") -> torch.Tensor:
    """[B, N, C] -> [B, C] fp32: 'mean' (x.mean(1)), 'max' (x.max(1)[0]) or 'cls' (x[:, 0])."""
    if mode not in _POOL:
        raise MetaEncError(f"pool_tokens: mode {mode!r} (mean / max / cls)")
    return _PoolFn.apply(x, _POOL[mode])


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on me_gemm (exact fp32 MFMA or bf16 MFMA by the input dtype); N padded to a multiple of 8 internally
    (class counts such as 174 are not)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        cdt = torch.bfloat16 if x2.dtype in (torch.bfloat16, torch.float16) else torch.float32
        xc = x2 if x2.dtype == cdt else ops.cast(x2, cdt)
        N, K = w.shape
        Np = (N + 7) // 8 * 8
        wc = w.detach().to(cdt)
        bc = b.detach().float() if b is not None else None
        if Np != N:
            wc = torch.cat([wc, wc.new_zeros(Np - N, K)])
            bc = torch.cat([bc, bc.new_zeros(Np - N)]) if bc is not None else None
        y = ops.gemm(xc, wc.contiguous(), bias=bc, out_dtype=torch.float32)
        ctx.save_for_backward(xc, wc)
        ctx.meta = (N, Np, x.shape, b is not None, w.dtype, x.dtype)
        # never a VIEW of a tensor made in here: the reference's ClsHead puts nn.ReLU(inplace=True) straight behind its Linear
        # (cls_base.py:112-118 with norm_args=None), and autograd refuses in-place writes to a custom Function's view output
        return y if Np == N else y[:, :N].contiguous()

    @staticmethod
    def backward(ctx, dy):
        xc, wc = ctx.saved_tensors
        N, Np, xshape, has_b, wdt, xdt = ctx.meta
        d2 = dy.to(xc.dtype)
        if Np != N:
            d2 = torch.cat([d2, d2.new_zeros(d2.shape[0], Np - N)], dim=1)
        d2 = d2.contiguous()
        dx = ops.gemm(d2, ops.transpose_cast(wc, wc.dtype), out_dtype=torch.float32).reshape(xshape).to(xdt)
        M = d2.shape[0]
        if M % 8 == 0:
            dw = ops.gemm(d2, xc, op=_capi.ME_GEMM_TN, out_dtype=torch.float32)          # [Np, K]
        else:       # the TN kernel wants 16-byte rows of the reduction-major operands: pad the batch with zero rows
            pad = 8 - M % 8
            dw = ops.gemm(torch.cat([d2, d2.new_zeros(pad, Np)]), torch.cat([xc, xc.new_zeros(pad, xc.shape[1])]),
                          op=_capi.ME_GEMM_TN, out_dtype=torch.float32)
        db = ops.colsum(d2)[:N].to(wdt) if has_b else None
        return dx, dw[:N].to(wdt), db


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """F.linear on the HIP GEMM (fp32 result)."""
    if not x.is_cuda:
        raise MetaEncError("heads.linear: CUDA tensors only (no CPU fallback)")
    if x.shape[-1] % 8 != 0:
        raise MetaEncError(f"heads.linear: in_features {x.shape[-1]} must be a multiple of 8")
    y = _LinearFn.apply(x, weight, bias)                        # [rows, out_features]
    return y if x.dim() == 2 else y.reshape(*x.shape[:-1], weight.shape[0])


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = x.contiguous()
        y, mean, rstd = ops.layernorm_fwd(x, w, b, eps, torch.float32, save_stats=True)
        ctx.save_for_backward(x, mean, rstd, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, w = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(dy.float().contiguous(), x, mean, rstd, w, None, x.dtype, True)
        return dx, dg.to(w.dtype), db.to(w.dtype), None


class ClassifierHead(nn.Module):
    """The Video / Image classification tail (Video/models/modeling_finetune.py:395, 445-454): parameters ``fc_norm.*``
    (use_mean_pooling=True) or ``norm.*``, and ``head.*``."""

    def __init__(self, embed_dim=768, num_classes=400, use_mean_pooling=True, norm_layer=None, head_drop_rate=0.0, init_scale=0.0):
        super().__init__()
        from functools import partial
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.use_mean_pooling = use_mean_pooling
        if use_mean_pooling:
            self.fc_norm = norm_layer(embed_dim)
        else:
            self.norm = norm_layer(embed_dim)
        self.head_dropout = nn.Dropout(head_drop_rate)
        self.head = nn.Linear(embed_dim, num_classes)
        if init_scale:
            self.head.weight.data.mul_(init_scale)
            self.head.bias.data.mul_(init_scale)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ln = self.fc_norm if self.use_mean_pooling else self.norm
        # LayerNorm is per token, so norm(x)[:, 0] == norm(x[:, 0]): pool first, normalise B rows instead of B*N
        f = pool_tokens(x, "mean" if self.use_mean_pooling else "cls")
        f = _LayerNormFn.apply(f, ln.weight, ln.bias, ln.eps)
        return linear(self.head_dropout(f), self.head.weight, self.head.bias)


class ClsHead(nn.Module):
    """openpoints ClsHead (cls_base.py:77-136).  ``global_feat`` e.g. 'max,avg' pools the token axis and concatenates;
    blocks follow create_linearblock: nn.Sequential(Linear, [BatchNorm1d], [ReLU]) -> keys ``head.{i}.0.weight`` ..."""

    def __init__(self, num_classes: int, in_channels: int, mlps=(256,), norm_args=None, act_args=None, dropout: float = 0.5,
                 global_feat: Optional[str] = None, point_dim: int = 2):
        super().__init__()
        self.global_feat = global_feat.split(",") if global_feat is not None else None
        self.point_dim = point_dim
        act_args = {"act": "relu"} if act_args is None else act_args
        cin = len(self.global_feat) * in_channels if self.global_feat is not None else in_channels
        dims = [cin] + list(mlps or []) + [num_classes]
        heads = []
        for i in range(len(dims) - 2):
            blk = [nn.Linear(dims[i], dims[i + 1], bias=not (norm_args and "bn" in norm_args.get("norm", "")))]
            if norm_args and "bn" in norm_args.get("norm", ""):
                blk.append(nn.BatchNorm1d(dims[i + 1]))
            elif norm_args and "ln" in norm_args.get("norm", ""):
                blk.append(nn.LayerNorm(dims[i + 1]))
            if act_args:
                blk.append(nn.ReLU(inplace=True))
            heads.append(nn.Sequential(*blk))
            if dropout:
                heads.append(nn.Dropout(dropout))
        heads.append(nn.Sequential(nn.Linear(dims[-2], dims[-1])))
        self.head = nn.Sequential(*heads)

    def forward(self, end_points: torch.Tensor) -> torch.Tensor:
        f = end_points
        if self.global_feat is not None:
            if f.dim() != 3:
                raise MetaEncError("ClsHead(global_feat=...) expects 3-D features")
            if self.point_dim == 2:                 # the reference's channel-first [B, C, N] convention
                f = f.transpose(1, 2)
            f = torch.cat([pool_tokens(f.contiguous(), "max" if "max" in g else "mean") for g in self.global_feat], dim=1)
        for m in self.head:
            if isinstance(m, nn.Sequential):
                lin = m[0]
                f = linear(f, lin.weight, lin.bias)
                for sub in list(m)[1:]:
                    f = sub(f)
            else:
                f = m(f)
        return f


# ----------------------------------------------------------------------------------------------------- checkpoint I/O

_BLOCK_KEY = re.compile(r"(?:^|\.)(\d+\.(?:norm1|norm2|attn|mlp|gamma_?[12])\b.*)$")


def _strip(sd):
    """keep the encoder blocks' entries and drop whatever wraps them (module. / backbone. / encoder. / blocks. ...):
    'module.blocks.3.attn.qkv.weight' -> '3.attn.qkv.weight'; patch_embed / head / pos_embed entries are not the encoder's"""
    out = OrderedDict()
    for k, v in sd.items():
        m = _BLOCK_KEY.search(k)
        if m:
            out[m.group(1)] = v
    return out


def load_encoder_checkpoint(encoder: nn.Sequential, source, strict: bool = True, map_location="cpu") -> nn.Sequential:
    """``source``: a path to the reference's .pth (torch.load -> bare state dict or {'model' / 'state_dict': ...}) or a state
    dict.  Accepts the wire format of README.md:125-135, wrapper prefixes, the Video key set (q_bias / v_bias / gamma_1 /
    gamma_2) and fp16 / bf16 tensors; non-encoder keys (patch_embed, head, ...) are ignored unless strict demands every
    encoder key is present.  Loads with the reference's own ``load_state_dict(strict=True)`` call."""
    sd = torch.load(source, map_location=map_location, weights_only=True) if isinstance(source, (str, bytes)) or hasattr(source, "read") else source
    for k in ("model", "state_dict", "module"):
        if isinstance(sd, dict) and k in sd and isinstance(sd[k], dict):
            sd = sd[k]
    sd = convert_video_state_dict(_strip(OrderedDict(sd)))
    want = encoder.state_dict()
    sd = OrderedDict((k, v.to(want[k].dtype) if k in want else v) for k, v in sd.items())
    encoder.load_state_dict(sd, strict=strict)
    return encoder


def save_encoder_checkpoint(encoder: nn.Sequential, path, dtype: torch.dtype = torch.float32) -> None:
    """Writes the reference wire format: a bare OrderedDict of CPU tensors under the timm key names."""
    sd = OrderedDict((k, v.detach().to("cpu", dtype)) for k, v in encoder.state_dict().items())
    torch.save(sd, path)


def pack_encoder(encoder: nn.Sequential, device=None, warm: bool = True):
    """Device-resident packed layout of an encoder's weights: returns a parallel.FlatParams whose flat fp32 buffer holds
    every parameter (each nn.Parameter becomes a view of it, no-decay parameters first) and, with ``warm``, builds the
    compute-dtype copies the kernels read (bf16 forward copies and transposed dgrad copies, one batched launch)."""
    from . import parallel
    if device is not None:
        encoder.to(device)
    if not all(isinstance(b, Block) for b in encoder):
        raise MetaEncError("pack_encoder: an nn.Sequential of Blocks is required")
    for p in encoder.parameters():
        p.requires_grad_(True)
    flat = parallel.FlatParams(encoder.named_parameters(), no_decay=parallel.no_decay_rule)
    if warm:
        for b in encoder:
            b._wcache.bind(b)
            # the dtype the Block will compute in (compute_dtype may be the STRING "fp32_3xbf16", which is not a torch dtype)
            cdt = torch.float32 if b.compute_dtype == "fp32_3xbf16" else (b.compute_dtype or torch.bfloat16)
            x3 = b.uses_3xbf16(cdt, torch.float32)          # fp32-accurate mode: its compute copies are the three-plane split weights
            for name, w in (("qkv", b.attn.qkv.weight), ("proj", b.attn.proj.weight), ("fc1", b.mlp.fc1.weight), ("fc2", b.mlp.fc2.weight)):
                if x3:
                    b._wcache.split3(name, w, False)
                    b._wcache.split3(name, w, True)
                else:
                    b._wcache.fwd(name, w, cdt)
                    b._wcache.transposed(name, w, cdt)
    return flat


# ----------------------------------------------------------------------------------------------------- point-cloud tokenizer

def furthest_point_sample(p: torch.Tensor, m: int) -> torch.Tensor:
    """[B, n, 3] fp32 -> [B, m] int32 indices (me_fps; idx[:, 0] = 0 as in the reference kernel)."""
    if p.dim() != 3 or p.shape[2] != 3 or not p.is_cuda:
        raise MetaEncError("furthest_point_sample: [B, n, 3] CUDA points required (no CPU fallback)")
    p = p.float().contiguous()
    B, n, _ = p.shape
    idx = torch.empty(B, m, dtype=torch.int32, device=p.device)
    tmp = torch.empty(B, n, dtype=torch.float32, device=p.device)
    check(_capi.load().me_fps(ptr(p), ptr(idx), ptr(tmp), B, n, m, stream_ptr()), "me_fps")
    return idx


def knn_indices(support: torch.Tensor, query: torch.Tensor, k: int) -> torch.Tensor:
    """[B, n, 3], [B, m, 3] -> [B, m, k] int32: the k nearest support points of each query, nearest first (me_knn)."""
    support, query = support.float().contiguous(), query.float().contiguous()
    B, n, _ = support.shape
    m = query.shape[1]
    idx = torch.empty(B, m, k, dtype=torch.int32, device=support.device)
    check(_capi.load().me_knn(ptr(support), ptr(query), ptr(idx), B, n, m, k, stream_ptr()), "me_knn")
    return idx


class PointPatchEmbed(nn.Module):
    """openpoints PointPatchEmbed (group_embed.py:60-172) for the configuration the Meta-Transformer point-cloud pipelines
    use (cfgs/modelnet40ply2048/metatransformer.yaml:19-33): FPS subsampling, KNN grouping, feature_type 'dp' (relative
    xyz), conv-norm-act blocks with BatchNorm2d + ReLU, max reduction.  Parameter names follow the reference's
    ``conv1.{i}.0.weight`` ([out, in, 1, 1] Conv2d) / ``conv1.{i}.1.*`` (BatchNorm2d) layout, so its checkpoints load.

    forward(p) returns the reference's ``[p, center_p], [x, out_f]`` with out_f channel-first [B, C, S]; ``tokens(p)`` gives
    the [B, S, C] token layout the encoder consumes.  FPS / KNN / grouping, every 1x1 convolution (a GEMM over B*S*k rows)
    and the max reductions run in libmetaenc.so; BatchNorm / ReLU are PyTorch glue."""

    def __init__(self, sample_ratio=0.25, group_size=32, in_channels=3, layers=4, embed_dim=768, channels=(128, 256, 512),
                 subsample="fps", group="knn", feature_type="dp", norm_args=None, reduction="max"):
        super().__init__()
        if subsample != "fps" or "knn" not in group or feature_type != "dp" or reduction != "max":
            raise MetaEncError("PointPatchEmbed: fps + knn + feature_type 'dp' + max reduction is the implemented configuration")
        self.sample_ratio, self.group_size = sample_ratio, group_size
        ch = [3] + list(channels) + [embed_dim]
        layers = len(ch) - 1
        half = layers // 2

        def block(cin, cout, last):
            mods = [nn.Conv2d(cin, cout, 1, bias=last)]
            if not last:
                mods += [nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]
            return nn.Sequential(*mods)
        self.conv1 = nn.Sequential(*[block(ch[i], ch[i + 1], i == half - 1) for i in range(half)])
        ch2 = list(ch)
        ch2[half] *= 2
        self.conv2 = nn.Sequential(*[block(ch2[i], ch2[i + 1], i == layers - 1) for i in range(half, layers)])
        self.out_channels = ch[-1]

    def _mlp(self, seq: nn.Sequential, f: torch.Tensor) -> torch.Tensor:
        for blk in seq:
            conv = blk[0]
            w = conv.weight.reshape(conv.out_channels, conv.in_channels)
            K = f.shape[1]
            if w.shape[1] != K:                                   # the xyz layer: reduction padded 3 -> 8
                w = torch.cat([w, w.new_zeros(w.shape[0], K - w.shape[1])], dim=1)
            f = linear(f, w, conv.bias)
            if len(blk) > 1:
                bn = blk[1]
                f = torch.nn.functional.batch_norm(f, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, bn.momentum, bn.eps)
                f = torch.relu_(f)
        return f

    def tokens(self, p: torch.Tensor):
        B, n, _ = p.shape
        S, k = int(n * self.sample_ratio), self.group_size
        p = p.float().contiguous()
        idx = furthest_point_sample(p, S)
        center = torch.gather(p, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        nbr = knn_indices(p, center, k)
        rows = torch.empty(B * S * k, 8, dtype=torch.float32, device=p.device)
        check(_capi.load().me_group_relative(ptr(p), ptr(center), ptr(nbr), ptr(rows), B, n, S, k, 8, stream_ptr()), "me_group_relative")
        f = self._mlp(self.conv1, rows)                                               # [B*S*k, C1]
        C1 = f.shape[1]
        pooled = pool_tokens(f.reshape(B * S, k, C1), "max")                          # [B*S, C1]
        f = torch.cat([pooled.unsqueeze(1).expand(-1, k, -1), f.reshape(B * S, k, C1)], dim=2).reshape(B * S * k, 2 * C1)
        f = self._mlp(self.conv2, f.contiguous())
        out = pool_tokens(f.reshape(B * S, k, f.shape[1]), "max").reshape(B, S, -1)
        return out, center, idx, nbr

    def forward(self, p: torch.Tensor, x: Optional[torch.Tensor] = None):
        out, center, _, _ = self.tokens(p)
        return [p, center], [x, out.transpose(1, 2)]
