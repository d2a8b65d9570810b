"""Data2Seq tokenizers on the HIP path: ``Data2Seq(modality, dim)(data) -> [B, N, dim]``.

Mirrors the reference's tokenizer plugin point (Data2Seq/Data2Seq.py:19-55, README.md:113-122) for the
modalities whose tokenizers are on the north-star path:

  image        Data2Seq/Image.py:8-28        Conv2d(3, C, k16, s16)           -> MFMA GEMM gathering its own patches
  audio        Data2Seq/Acoustic.py:5-23     Conv2d(1, C, k16, stride 10x10)  -> overlapping gather + GEMM
               (as written the reference class cannot be constructed -- it double-tuples patch_size; the working
               construction is Audio/src/models/ast_models.py:86, which this follows)
  video        Video/models/modeling_finetune.py:263-297  Conv3d(3, C, k=s=(2,16,16)) tubelets
               (Data2Seq/Video.py is broken in the reference)
  time-series  Data2Seq/Time_Series.py:109-126  Conv1d k3 circular + sinusoid PE + temporal-embedding gathers

Parameter names follow the reference modules (``proj.weight`` / ``proj.bias``;
``value_embedding.tokenConv.weight`` ...), so their state_dicts interchange.  The text / graph / hyper-spectral
tokenizers of the reference are CLIP / eigen-decomposition / broken glue and are out of scope.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import torch
import torch.nn as nn

from . import _capi, ops
from ._capi import ME_GEMM_TN, MetaEncError, check, dtype_code, ptr, stream_ptr


def _boundary_dtypes(weight: torch.Tensor):
    """(compute dtype, output dtype) at a tokenizer's boundary, the rule Block follows: fp32 parameters -> exact fp32
    kernels; bf16 (parameters or autocast) -> bf16 MFMA kernels; fp16 (``.half()`` models, ``torch.autocast(float16)`` --
    Audio/src/traintest.py) is a STORAGE dtype of the boundary only: tensors are converted by me_cast, the kernels compute
    in bf16 and the result goes back out as fp16."""
    dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else weight.dtype
    if dt == torch.float16:
        return torch.bfloat16, torch.float16
    if dt in (torch.bfloat16, torch.float32):
        return dt, dt
    raise MetaEncError(f"tokenizer dtype {dt} unsupported (fp32, bf16, fp16)")


class _PatchEmbedFn(torch.autograd.Function):
    """conv-as-GEMM through me_patch_embed: for bf16 compute on the reference's image / tubelet geometries the patch gather runs
    inside the MFMA GEMM's operand stager (no gathered matrix in memory); bias, optional pos-embed add and the cls-token row offset
    are the GEMM's epilogue.  Backward: me_patch_embed_wgrad gathers the same patches inside the weight-gradient kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, pos, geom, cdt, prefix_rows, odt=None):
        kt, kh, kw, st, sh, sw = geom
        B = x.shape[0]
        Cout = weight.shape[0]
        x_dtype = x.dtype
        x = x.contiguous()
        if x.dtype == torch.float16:                      # fp16 is converted at the boundary (me_cast), never computed in
            x = ops.cast(x, torch.bfloat16)
        if cdt == torch.bfloat16 and x.dtype == torch.float32 and ops.patch_embed_fused(x, geom, cdt, Cout, x_dtype=cdt):
            # the gather rounds the pixels to bf16 either way: one vectorised cast pass, then the fused kernel reads them where they lie
            x = ops.cast(x, cdt)
        w2 = ops.cast(weight.detach().reshape(Cout, -1).contiguous(), cdt)
        pos2 = None
        if pos is not None:
            pos2 = pos.detach().reshape(-1, Cout)
        y, tps = ops.patch_embed(x, w2, bias, pos2, geom, prefix_rows, cdt)
        out_tps = tps + prefix_rows
        ctx.save_for_backward(x, weight)
        ctx.meta = (geom, cdt, tps, out_tps, prefix_rows, bias is not None, pos is not None, x_dtype)
        if odt is not None and odt != y.dtype:
            y = ops.cast(y, odt)
        return y.reshape(B, out_tps, Cout)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        geom, cdt, tps, out_tps, prefix_rows, has_bias, has_pos, x_dtype = ctx.meta
        kt, kh, kw, st, sh, sw = geom
        B = x.shape[0]
        Cout = weight.shape[0]
        dy = dy.contiguous()
        if prefix_rows:
            dy = dy[:, prefix_rows:, :].contiguous()
        dy2 = ops.cast(dy.reshape(B * tps, Cout), cdt)
        ng = ctx.needs_input_grad
        dW = db = dx = dpos = None
        want_db = bool(ng[2] and has_bias)
        if ng[1]:
            wdt = torch.float32 if weight.dtype == torch.float16 else weight.dtype
            dW, db = ops.patch_embed_wgrad(x, geom, dy2, wdt, want_db)
            dW = ops.cast(dW, weight.dtype).reshape(weight.shape)
            if db is not None:
                db = db.to(weight.dtype)
        elif want_db:
            db = ops.colsum(dy2).to(weight.dtype)
        if ng[0]:
            wT = ops.transpose_cast(weight.detach().reshape(Cout, -1).contiguous(), cdt)     # [K, Cout]
            dcols = ops.gemm(dy2, wT)                                                        # [B*tps, K]
            dx = ops.unpatchify_add(dcols, tuple(x.shape), kt, kh, kw, st, sh, sw)
            if dx.dtype != x_dtype:
                dx = ops.cast(dx, x_dtype)
        if has_pos and ng[3]:
            raise MetaEncError("gradient w.r.t. a fused pos-embed is not implemented; add it outside the tokenizer")
        return dx, dW, db, dpos, None, None, None, None


class _ConvPatchEmbed(nn.Module):
    geom = (1, 16, 16, 1, 16, 16)

    def forward(self, x: torch.Tensor, pos_embed: Optional[torch.Tensor] = None, prefix_rows: int = 0) -> torch.Tensor:
        if not x.is_cuda:
            raise MetaEncError(f"{type(self).__name__} runs on MI355X only (CPU tensor given; no CPU fallback)")
        self._check_input(x)
        cdt, odt = _boundary_dtypes(self.proj.weight)
        return _PatchEmbedFn.apply(x, self.proj.weight, self.proj.bias, pos_embed, self.geom, cdt, prefix_rows, odt)

    def _check_input(self, x):
        pass


class PatchEmbed(_ConvPatchEmbed):
    """2D Image to Patch Embedding -- Data2Seq/Image.py:4-28 (same constructor, same parameter names)."""

    def __init__(self, img_size=224, patch_size=16, in_c=3, embed_dim=768, norm_layer=None):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_c, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        if norm_layer is not None:
            raise MetaEncError("norm_layer is unused by the reference forward (Data2Seq/Image.py:27) and unsupported")
        self.norm = nn.Identity()
        self.geom = (1, patch_size, patch_size, 1, patch_size, patch_size)

    def _check_input(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."


class AcousticPatchEmbed(_ConvPatchEmbed):
    """Spectrogram patch embed, Conv2d(in_chans, C, k=(16,16), stride=(fstride,tstride)) -- Data2Seq/Acoustic.py:5-23,
    Audio/src/models/ast_models.py:86."""

    def __init__(self, img_size=224, patch_size=16, in_chans=1, embed_dim=768, fstride=10, tstride=10):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=(fstride, tstride))
        self.geom = (1, patch_size, patch_size, 1, fstride, tstride)

    @staticmethod
    def num_tokens(F: int, T: int, patch: int = 16, fstride: int = 10, tstride: int = 10) -> int:
        return ((F - patch) // fstride + 1) * ((T - patch) // tstride + 1)


class VideoPatchEmbed(_ConvPatchEmbed):
    """Tubelet embed, Conv3d(3, C, k=s=(tubelet,16,16)) -- Video/models/modeling_finetune.py:263-297."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, num_frames=16, tubelet_size=2):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.tubelet_size = tubelet_size
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2 * (num_frames // tubelet_size)
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=(tubelet_size, patch_size, patch_size),
                              stride=(tubelet_size, patch_size, patch_size))
        self.geom = (tubelet_size, patch_size, patch_size, tubelet_size, patch_size, patch_size)

    def _check_input(self, x):
        B, C, T, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."


def sinusoid_table(n: int, d_model: int) -> torch.Tensor:
    """PositionalEmbedding / FixedEmbedding table -- Data2Seq/Time_Series.py:12-23,49-57 (host-side constant)."""
    pe = torch.zeros(n, d_model).float()
    position = torch.arange(0, n).float().unsqueeze(1)
    div_term = (torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model)).exp()
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def video_sinusoid_table(n_position: int, d_hid: int) -> torch.Tensor:
    """get_sinusoid_encoding_table -- Video/models/modeling_finetune.py:302-318 (float64 math, float32 result)."""
    pos = torch.arange(n_position, dtype=torch.float64).unsqueeze(1)
    j = torch.arange(d_hid, dtype=torch.float64)
    angle = pos / torch.pow(torch.tensor(10000.0, dtype=torch.float64), 2 * torch.div(j, 2, rounding_mode="floor") / d_hid)
    tab = angle.clone()
    tab[:, 0::2] = torch.sin(angle[:, 0::2])
    tab[:, 1::2] = torch.cos(angle[:, 1::2])
    return tab.float().unsqueeze(0)


class _TokenConv(nn.Module):
    def __init__(self, c_in, d_model):
        super().__init__()
        self.tokenConv = nn.Conv1d(c_in, d_model, kernel_size=3, padding=1, padding_mode="circular", bias=False)
        nn.init.kaiming_normal_(self.tokenConv.weight, mode="fan_in", nonlinearity="leaky_relu")


class _FixedEmb(nn.Module):
    def __init__(self, c_in, d_model):
        super().__init__()
        self.emb = nn.Embedding(c_in, d_model)
        self.emb.weight = nn.Parameter(sinusoid_table(c_in, d_model), requires_grad=False)


class _Temporal(nn.Module):
    SIZES = (("month", 13), ("day", 32), ("weekday", 7), ("hour", 24), ("minute", 4))   # mark column order 0..4

    def __init__(self, d_model, freq="h"):
        super().__init__()
        if freq == "t":
            self.minute_embed = _FixedEmb(4, d_model)
        self.hour_embed = _FixedEmb(24, d_model)
        self.weekday_embed = _FixedEmb(7, d_model)
        self.day_embed = _FixedEmb(32, d_model)
        self.month_embed = _FixedEmb(13, d_model)

    def tables(self):
        names = ["month", "day", "weekday", "hour"] + (["minute"] if hasattr(self, "minute_embed") else [])
        return [getattr(self, f"{n}_embed").emb.weight for n in names]


class _PosEmb(nn.Module):
    def __init__(self, d_model, max_len=5000):
        super().__init__()
        self.register_buffer("pe", sinusoid_table(max_len, d_model).unsqueeze(0))


class DataEmbedding(nn.Module):
    """Time-series DataEmbedding -- Data2Seq/Time_Series.py:109-126 (``embed_type='fixed'``).
    forward(x [B,L,c_in], x_mark [B,L,4|5] or None) -> [B,L,C].  The three terms (circular Conv1d k3, temporal
    table gathers indexed by ``x_mark.long()``, positional slice ``pe[:, :L]``) are one fused kernel; the gathers
    are integer-indexed and bit-exact.  Dropout(p=0.1) is identity in eval; training-mode dropout is applied by
    the caller's nn.Dropout if wanted."""

    def __init__(self, c_in, d_model, embed_type="fixed", freq="h", dropout=0.1):
        super().__init__()
        if embed_type != "fixed":
            raise MetaEncError("only embed_type='fixed' (the reference default) is implemented")
        self.value_embedding = _TokenConv(c_in, d_model)
        self.position_embedding = _PosEmb(d_model)
        self.temporal_embedding = _Temporal(d_model, freq)
        self.dropout = nn.Dropout(p=dropout)
        self.d_model = d_model

    def forward(self, x: torch.Tensor, x_mark: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not x.is_cuda:
            raise MetaEncError("DataEmbedding runs on MI355X only (CPU tensor given; no CPU fallback)")
        if x.requires_grad:
            raise MetaEncError("DataEmbedding: gradient w.r.t. the input series is not implemented (no reference pipeline "
                               "needs it); detach the input")
        B, L, cin = x.shape
        pe = self.position_embedding.pe[0]
        if L > pe.shape[0]:
            raise MetaEncError(f"sequence length {L} exceeds positional table {pe.shape[0]}")
        marks, tabs = None, []
        if x_mark is not None:
            tabs = [t.detach().float().contiguous() for t in self.temporal_embedding.tables()]
            if x_mark.shape[-1] < len(tabs):
                raise MetaEncError(f"x_mark has {x_mark.shape[-1]} columns, need {len(tabs)}")
            if x_mark.device != x.device:
                raise MetaEncError(f"x_mark is on {x_mark.device}, the series on {x.device}: move it first (no implicit copies; "
                                   "a host pointer handed to the kernel would fault the GPU)")
            marks = x_mark[..., :len(tabs)].long().to(torch.int32).contiguous()      # == x.long() (Time_Series.py:83)
        acast = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else None
        if acast not in (None, torch.bfloat16, torch.float16, torch.float32):
            raise MetaEncError(f"autocast dtype {acast} unsupported")
        out_dtype = torch.bfloat16 if acast in (torch.bfloat16, torch.float16) else torch.float32      # fp16: bf16 compute, cast out
        p = float(self.dropout.p) if self.training else 0.0
        seed = int(torch.empty((), dtype=torch.int64).random_().item()) if p > 0 else 0
        y = _TSEmbedFn.apply(x.detach().float().contiguous(), self.value_embedding.tokenConv.weight, marks, tabs,
                             pe.contiguous(), out_dtype, p, seed)
        return y.to(torch.float16) if acast == torch.float16 else y


class _TSEmbedFn(torch.autograd.Function):
    """DataEmbedding.forward (Data2Seq/Time_Series.py:118-126): value + temporal + positional embedding, Dropout(p).
    Backward: the only trainable tensor of the default (embed_type='fixed') configuration is the Conv1d weight;
    dW[C, cin, 3] = dY^T unfold(x) as one TN GEMM on the circularly unfolded input (me_timeseries_unfold)."""

    @staticmethod
    def forward(ctx, xf, weight, marks, tabs, pe, out_dtype, p, seed):
        lib = _capi.load()
        B, L, cin = xf.shape
        C = weight.shape[0]
        w = weight.detach().float().contiguous()
        out = torch.empty((B, L, C), dtype=out_dtype, device=xf.device)
        err = torch.zeros(1, dtype=torch.int32, device=xf.device)
        n_mark, tab_arr, rows_arr = len(tabs), None, None
        if marks is not None:
            tab_arr = (ctypes.c_void_p * n_mark)(*[t.data_ptr() for t in tabs])
            rows_arr = (ctypes.c_int32 * n_mark)(*[t.shape[0] for t in tabs])
        check(lib.me_timeseries_embed(ptr(xf), ptr(w), ptr(marks), n_mark if marks is not None else 0, tab_arr, rows_arr, ptr(pe),
                                      ptr(out), dtype_code(out_dtype), B, L, cin, C, ptr(err), stream_ptr()),
              "me_timeseries_embed")
        if marks is not None and int(err.item()) != 0:
            raise IndexError("x_mark holds an index outside its embedding table (nn.Embedding would raise too)")
        if p > 0:      # nn.Dropout(p) of Time_Series.py:126, training mode
            out = ops.dropout_add(out, None, L, p, 0.0, seed)
        ctx.save_for_backward(xf)
        ctx.meta = (p, seed, weight.dtype, weight.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        (xf,) = ctx.saved_tensors
        p, seed, wdt, wshape = ctx.meta
        if not ctx.needs_input_grad[1]:
            return (None,) * 8
        lib = _capi.load()
        B, L, cin = xf.shape
        C = wshape[0]
        dy2 = dy.contiguous()
        if p > 0:
            dy2 = ops.dropout_add(dy2, None, L, p, 0.0, seed)
        dy2 = ops.cast(dy2.reshape(B * L, C), torch.float32)
        ncols = (3 * cin + 3) // 4 * 4
        xu = torch.empty((B * L, ncols), dtype=torch.float32, device=xf.device)
        check(lib.me_timeseries_unfold(ptr(xf), ptr(xu), B, L, cin, ncols, stream_ptr()), "me_timeseries_unfold")
        dw = ops.gemm(dy2, xu, op=_capi.ME_GEMM_TN, out_dtype=torch.float32)          # [C, ncols]
        dw = dw[:, :3 * cin].reshape(C, cin, 3).to(wdt)
        return (None, dw, None, None, None, None, None, None)


class Data2Seq(nn.Module):
    """``Data2Seq(modality, dim)`` dispatcher -- API shape of Data2Seq/Data2Seq.py:19-55 (which itself does not run
    as written: SURVEY.md appendix A).  Multi-modal use concatenates along tokens exactly as README.md:118-122:
    ``features = torch.concat([image_tokenizer(img), ts_tokenizer(ts), audio_tokenizer(spec)], dim=1)``."""

    def __init__(self, modality: str, dim: int, **kw):
        super().__init__()
        self.modality = modality
        if modality == "image":
            self.embed = PatchEmbed(embed_dim=dim, **kw)
        elif modality == "audio":
            self.embed = AcousticPatchEmbed(embed_dim=dim, **kw)
        elif modality == "video":
            self.embed = VideoPatchEmbed(embed_dim=dim, **kw)
        elif modality == "time-series":
            self.embed = DataEmbedding(c_in=kw.pop("c_in", 1), d_model=dim, **kw)
        else:
            raise MetaEncError(f"modality '{modality}' has no HIP tokenizer (in scope: image, audio, video, time-series)")

    def forward(self, data, *args, **kw):
        return self.embed(data, *args, **kw)
