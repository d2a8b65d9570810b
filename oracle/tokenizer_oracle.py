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


def _cubic_weights(t: torch.Tensor, A: float = -0.75):
    """ATen's cubic-convolution coefficients (UpSample.h get_cubic_upsample_coefficients)."""
    def c1(x): return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0
    def c2(x): return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A
    return [c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)]


def _resize_axis_taps(n_in: int, n_out: int, mode: str):
    """per output index: (tap indices [n_out, T], tap weights [n_out, T]); align_corners=False"""
    scale = n_in / n_out
    d = torch.arange(n_out, dtype=torch.float32)
    s = scale * (d + 0.5) - 0.5
    if mode == "bicubic":
        fl = torch.floor(s)
        t = s - fl
        idx = torch.stack([fl.long() + k for k in (-1, 0, 1, 2)], dim=1).clamp(0, n_in - 1)
        return idx, torch.stack(_cubic_weights(t), dim=1)
    s = s.clamp_min(0.0)
    i0 = s.long()
    i1 = torch.where(i0 < n_in - 1, i0 + 1, i0)
    lam = s - i0.float()
    return torch.stack([i0, i1], dim=1), torch.stack([1.0 - lam, lam], dim=1)


def resize_pos_embed(pos_embed: torch.Tensor, input_shape, pos_shape, mode: str = "bicubic") -> torch.Tensor:
    """TIMMVisionTransformer.resize_pos_embed, Image/detection/mmdet_custom/models/backbones/base/vit.py:459-486: keep row 0
    (cls), resample the last pos_h * pos_w rows as a [pos_h, pos_w] grid to input_shape with F.interpolate(mode,
    align_corners=False) (:478-479), restated as explicit separable taps (x first, then y, as ATen evaluates it)."""
    ph, pw = pos_shape
    H, W = input_shape
    C = pos_embed.shape[2]
    grid = pos_embed[0, -ph * pw:].reshape(ph, pw, C)
    ix, wx = _resize_axis_taps(pw, W, mode)
    iy, wy = _resize_axis_taps(ph, H, mode)
    rows = (grid[:, ix] * wx[None, :, :, None]).sum(dim=2)          # [ph, W, C]
    out = (rows[iy] * wy[:, :, None, None]).sum(dim=1)              # [H, W, C]
    return torch.cat([pos_embed[:, :1], out.reshape(1, H * W, C)], dim=1)


# ---- point-cloud tokenizer front end (numpy / torch-CPU restatements)

def fps_reference(points, m: int):
    """furthest_point_sampling_kernel, PointCloud/openpoints/cpp/pointnet2_batch/src/sampling_gpu.cu:101-210, with its
    thread-strided scan and binary-tree fold restated literally (T = largest power of two <= n, capped at 1024:
    opt_n_threads): start at index 0; temp = 1e10 (the caller's fill); each round every thread keeps the first strict
    maximum of min(d, temp[k]) over its points, partners (t, t + s) fold with `v2 > v1 ? i2 : i1`."""
    import numpy as np
    pts = np.asarray(points, dtype=np.float32)
    B, n, _ = pts.shape
    T = 1
    while T * 2 <= n and T < 1024:
        T *= 2
    out = np.zeros((B, m), dtype=np.int32)
    for b in range(B):
        P = pts[b]
        temp = np.full(n, 1e10, dtype=np.float32)
        old = 0
        for j in range(1, m):
            d = P - P[old]
            # the reference's expression (x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1) as hipcc evaluates it on gfx950
            # (ISA of oracle/_ref/libref_fps.so, all block sizes): fma(dy, dy, dx*dx) + dz*dz.  Evaluated in double and rounded
            # after each step (products of two floats are exact there)
            t0 = (d[:, 0].astype(np.float64) * d[:, 0]).astype(np.float32)
            t1 = (d[:, 1].astype(np.float64) * d[:, 1] + t0).astype(np.float32)
            t2 = (d[:, 2].astype(np.float64) * d[:, 2]).astype(np.float32)
            dist = (t1 + t2).astype(np.float32)
            temp = np.minimum(dist, temp)
            best = np.full(T, -1.0, dtype=np.float32)
            besti = np.zeros(T, dtype=np.int64)
            for t in range(min(T, n)):
                sub = temp[t::T]
                a = int(np.argmax(sub))                      # first maximum of the strided walk
                if sub[a] > -1.0:
                    best[t], besti[t] = sub[a], t + a * T
            s = T // 2
            while s >= 1:
                v1, v2 = best[:s].copy(), best[s:2 * s].copy()
                i1, i2 = besti[:s].copy(), besti[s:2 * s].copy()
                best[:s] = np.maximum(v1, v2)
                besti[:s] = np.where(v2 > v1, i2, i1)
                s //= 2
            old = int(besti[0])
            out[b, j] = old
    return out


def knn_reference(support: torch.Tensor, query: torch.Tensor, k: int):
    """KNN.forward, PointCloud/openpoints/models/layers/group.py:17-28: cdist + topk(largest=False); returns [B, m, k]."""
    dist = torch.cdist(support, query)
    return dist.topk(k=k, dim=1, largest=False).indices.transpose(1, 2).contiguous().int()


def point_patch_embed_reference(p: torch.Tensor, mod, fps_idx=None):
    """PointPatchEmbed.forward (group_embed.py:138-172) for feature_type 'dp', knn grouping, max reduction, on the module's own
    conv1 / conv2 (eval mode): returns out_f [B, C, S]."""
    B, n, _ = p.shape
    S, k = int(n * mod.sample_ratio), mod.group_size
    idx = torch.from_numpy(fps_reference(p.numpy(), S)).long() if fps_idx is None else fps_idx.long()
    center = torch.gather(p, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
    nbr = knn_reference(p, center, k).long()                                   # [B, S, k]
    grouped = torch.gather(p.unsqueeze(1).expand(-1, S, -1, -1), 2, nbr.unsqueeze(-1).expand(-1, -1, -1, 3))   # [B, S, k, 3]
    dp = (grouped - center.unsqueeze(2)).permute(0, 3, 1, 2)                   # [B, 3, S, k]  (relative_xyz, group.py:312-313)
    fj = mod.conv1(dp)
    fj = torch.cat([fj.max(dim=-1, keepdim=True)[0].expand(-1, -1, -1, k), fj], dim=1)
    return mod.conv2(fj).max(dim=-1)[0], idx, nbr
