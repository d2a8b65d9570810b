"""CPU fp32 restatement of the Data2Seq tokenizers that feed the encoder
(TEST INFRASTRUCTURE -- see oracle/__init__.py).  Paths relative to /root/reference.

The convolutions are restated as explicit patch gathers + matmuls so that the patch
index arithmetic (which must be bit-exact on the GPU) is visible and testable on its own.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch


# ----------------------------------------------------------------------------- patch index math

def patch_grid_2d(H: int, W: int, kh: int, kw: int, sh: int, sw: int) -> Tuple[int, int]:
    """Output grid of nn.Conv2d(k=(kh,kw), stride=(sh,sw), padding=0):
    Data2Seq/Image.py:16 (k=s=16) and Data2Seq/Acoustic.py:16 (k=16, stride=(10,10))."""
    return (H - kh) // sh + 1, (W - kw) // sw + 1


def patchify_2d(x: torch.Tensor, kh: int, kw: int, sh: int, sw: int) -> torch.Tensor:
    """[B,Cin,H,W] -> [B, gh*gw, Cin*kh*kw]; patch order row-major over (gh,gw) == conv output
    .flatten(2).transpose(1,2) (Data2Seq/Image.py:26); feature order (c,dy,dx) == Conv2d weight
    [Cout,Cin,kh,kw].reshape(Cout,-1)."""
    B, Cin, H, W = x.shape
    gh, gw = patch_grid_2d(H, W, kh, kw, sh, sw)
    out = x.new_empty(B, gh * gw, Cin * kh * kw)
    for py in range(gh):
        for px in range(gw):
            patch = x[:, :, py * sh:py * sh + kh, px * sw:px * sw + kw]
            out[:, py * gw + px] = patch.reshape(B, -1)
    return out


def patchify_3d(x: torch.Tensor, kt: int, kh: int, kw: int) -> torch.Tensor:
    """[B,Cin,T,H,W] -> [B, gt*gh*gw, Cin*kt*kh*kw] for the tubelet Conv3d with kernel == stride
    (Video/models/modeling_finetune.py:283-287); token order (t,h,w) row-major == flatten(2)."""
    B, Cin, T, H, W = x.shape
    gt, gh, gw = T // kt, H // kh, W // kw
    out = x.new_empty(B, gt * gh * gw, Cin * kt * kh * kw)
    for pt in range(gt):
        for py in range(gh):
            for px in range(gw):
                patch = x[:, :, pt * kt:(pt + 1) * kt, py * kh:(py + 1) * kh, px * kw:(px + 1) * kw]
                out[:, (pt * gh + py) * gw + px] = patch.reshape(B, -1)
    return out


# ----------------------------------------------------------------------------- tokenizers

def image_patch_embed(x, weight, bias, patch: int = 16):
    """Data2Seq/Image.py:19-28: Conv2d(in_c, C, k=patch, s=patch) -> flatten(2) -> transpose(1,2)."""
    cols = patchify_2d(x.float(), patch, patch, patch, patch)
    y = cols @ weight.reshape(weight.shape[0], -1).t().float()
    return y if bias is None else y + bias


def acoustic_patch_embed(x, weight, bias, fstride: int = 10, tstride: int = 10):
    """Data2Seq/Acoustic.py:16-23: Conv2d(1, C, k=(16,16), stride=(fstride,tstride)), overlapping."""
    kh, kw = weight.shape[-2:]
    cols = patchify_2d(x.float(), kh, kw, fstride, tstride)
    y = cols @ weight.reshape(weight.shape[0], -1).t().float()
    return y if bias is None else y + bias


def video_tubelet_embed(x, weight, bias):
    """Video/models/modeling_finetune.py:283-297: Conv3d(3, C, k=s=(tubelet,16,16))."""
    kt, kh, kw = weight.shape[-3:]
    cols = patchify_3d(x.float(), kt, kh, kw)
    y = cols @ weight.reshape(weight.shape[0], -1).t().float()
    return y if bias is None else y + bias


def sinusoid_table_ts(max_len: int, d_model: int) -> torch.Tensor:
    """PositionalEmbedding / FixedEmbedding table, Data2Seq/Time_Series.py:12-23,49-57."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len).float().unsqueeze(1)
    div_term = (torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model)).exp()
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def sinusoid_table_video(n_position: int, d_hid: int) -> torch.Tensor:
    """get_sinusoid_encoding_table, Video/models/modeling_finetune.py:302-318 (float64 math,
    cast to float32 at the end)."""
    pos = torch.arange(n_position, dtype=torch.float64).unsqueeze(1)
    j = torch.arange(d_hid, dtype=torch.float64)
    angle = pos / torch.pow(torch.tensor(10000.0, dtype=torch.float64), 2 * torch.div(j, 2, rounding_mode="floor") / d_hid)
    tab = angle.clone()
    tab[:, 0::2] = torch.sin(angle[:, 0::2])
    tab[:, 1::2] = torch.cos(angle[:, 1::2])
    return tab.float()


TEMPORAL_SIZES = {"month": 13, "day": 32, "weekday": 7, "hour": 24, "minute": 4}


def time_series_embedding(x: torch.Tensor, conv_weight: torch.Tensor,
                          x_mark: Optional[torch.Tensor] = None,
                          temporal_tables: Optional[Sequence[torch.Tensor]] = None,
                          pe: Optional[torch.Tensor] = None) -> torch.Tensor:
    """DataEmbedding.forward in eval mode, Data2Seq/Time_Series.py:109-126:

      value   = Conv1d(c_in, C, k=3, padding=1, padding_mode='circular', bias=False)  (:29-42)
                applied along L:  out[b,l,:] = sum_{j=0..2} W[:,:,j] @ x[b,(l-1+j) mod L,:]
      pos     = pe[:, :L]                                                            (:25-26)
      temporal= hour[x_mark[...,3]] + weekday[...,2] + day[...,1] + month[...,0] (+ minute[...,4])  (:82-93)
      out     = value + temporal + pos   (dropout p=0.1 is identity in eval)         (:118-126)

    ``temporal_tables`` = (month, day, weekday, hour[, minute]) tables, indexed by x_mark.long() column
    0,1,2,3[,4] -- the gathers are integer-indexed and must be bit-exact on the GPU.
    """
    B, L, cin = x.shape
    xf = x.float()
    C = conv_weight.shape[0]
    val = torch.zeros(B, L, C)
    for j in range(3):
        idx = (torch.arange(L) - 1 + j) % L
        val = val + xf[:, idx, :] @ conv_weight[:, :, j].t().float()
    out = val
    if x_mark is not None:
        m = x_mark.long()
        temporal = torch.zeros(B, L, C)
        # reference order of additions: hour + weekday + day + month + minute (:93)
        order = [3, 2, 1, 0] + ([4] if len(temporal_tables) > 4 else [])
        for col in order:
            temporal = temporal + temporal_tables[col][m[:, :, col]]
        out = out + temporal
    if pe is not None:
        out = out + pe[:L].unsqueeze(0)
    return out
