"""CPU fp32 restatement of the encoder hot path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Every function cites the reference lines it restates.  Paths are relative to
/root/reference.  The arithmetic is written out explicitly (matmul / exp / erf /
mean) instead of calling nn.LayerNorm / nn.GELU / F.softmax so that the oracle is
a restatement of the *algorithm*, and is validated against the reference's own
module in tests/test_oracle.py (via the committed golden vectors) and, in the
build container, directly (oracle/make_golden.py --check).

State-dict layout restated (keys verified against the in-tree Block):
    {i}.norm1.weight[C] {i}.norm1.bias[C]
    {i}.attn.qkv.weight[3C,C] {i}.attn.qkv.bias[3C]
    {i}.attn.proj.weight[C,C] {i}.attn.proj.bias[C]
    {i}.norm2.weight[C] {i}.norm2.bias[C]
    {i}.mlp.fc1.weight[4C,C] {i}.mlp.fc1.bias[4C]
    {i}.mlp.fc2.weight[C,4C] {i}.mlp.fc2.bias[C]
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional

import torch

BLOCK_KEYS = (
    "norm1.weight", "norm1.bias",
    "attn.qkv.weight", "attn.qkv.bias",
    "attn.proj.weight", "attn.proj.bias",
    "norm2.weight", "norm2.bias",
    "mlp.fc1.weight", "mlp.fc1.bias",
    "mlp.fc2.weight", "mlp.fc2.bias",
)


def block_param_shapes(dim: int, mlp_ratio: float = 4.0) -> "OrderedDict[str, tuple]":
    """Shapes of one Block's parameters, in state_dict order.
    Restates the constructors at PointCloud/openpoints/models/layers/attention.py:13-24,43-53
    and mlp.py:15-27 (nn.Linear weight is [out, in])."""
    hid = int(dim * mlp_ratio)
    return OrderedDict([
        ("norm1.weight", (dim,)), ("norm1.bias", (dim,)),
        ("attn.qkv.weight", (3 * dim, dim)), ("attn.qkv.bias", (3 * dim,)),
        ("attn.proj.weight", (dim, dim)), ("attn.proj.bias", (dim,)),
        ("norm2.weight", (dim,)), ("norm2.bias", (dim,)),
        ("mlp.fc1.weight", (hid, dim)), ("mlp.fc1.bias", (hid,)),
        ("mlp.fc2.weight", (dim, hid)), ("mlp.fc2.bias", (dim,)),
    ])


def make_encoder_state_dict(depth: int, dim: int, mlp_ratio: float = 4.0, seed: int = 0,
                            w_std: float = 0.02) -> "OrderedDict[str, torch.Tensor]":
    """Seeded random weights in the reference checkpoint layout (the real .pth files are
    Google-Drive downloads, README.md:101-104, not available).  LayerNorm affine and all
    biases are randomised too so that parity tests are sensitive to every parameter."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for i in range(depth):
        for name, shape in block_param_shapes(dim, mlp_ratio).items():
            if name.startswith("norm") and name.endswith("weight"):
                t = 1.0 + 0.1 * torch.randn(shape, generator=g)
            elif name.endswith("bias"):
                t = 0.05 * torch.randn(shape, generator=g)
            else:
                t = w_std * torch.randn(shape, generator=g)
            sd[f"{i}.{name}"] = t.float()
    return sd


def state_dict_checksum(sd: Dict[str, torch.Tensor]) -> float:
    """Order-sensitive checksum used by the golden files to pin the weight generator."""
    acc = 0.0
    for j, (k, v) in enumerate(sd.items()):
        acc += float(v.double().abs().sum()) * (1.0 + 1e-3 * (j % 97))
    return acc


# ----------------------------------------------------------------------------- ops

def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float) -> torch.Tensor:
    """nn.LayerNorm(C, eps) over the last dim (create_norm -> nn.LayerNorm,
    PointCloud/openpoints/models/layers/norm.py:65,97): biased variance, affine."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * weight + bias


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    """nn.GELU() exact-erf form (create_act('gelu') -> nn.GELU,
    PointCloud/openpoints/models/layers/activation.py:17,48-51)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """nn.Linear: y = x @ W^T + b with W [out, in]."""
    y = x @ w.t()
    return y if b is None else y + b


def attention(x: torch.Tensor, p: Dict[str, torch.Tensor], num_heads: int, prefix: str = "attn.",
              pre_scale_q: bool = False) -> torch.Tensor:
    """Attention.forward, PointCloud/openpoints/models/layers/attention.py:26-38.

    qkv = Linear(C,3C)(x).reshape(B,N,3,H,hd).permute(2,0,3,1,4)     (:28)
    attn = (q @ k^T) * scale ; softmax(-1)                            (:31-32)
    x = (attn @ v).transpose(1,2).reshape(B,N,C) ; proj               (:35-36)
    scale = head_dim ** -0.5                                          (:19)

    ``pre_scale_q`` restates the Video variant (q = q*scale before QK^T,
    Video/models/modeling_finetune.py:183-184); mathematically identical, different rounding.
    """
    B, N, C = x.shape
    hd = C // num_heads
    scale = hd ** -0.5
    qkv = linear(x, p[prefix + "qkv.weight"], p.get(prefix + "qkv.bias"))
    qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    if pre_scale_q:
        s = (q * scale) @ k.transpose(-2, -1)
    else:
        s = (q @ k.transpose(-2, -1)) * scale
    s = s - s.max(dim=-1, keepdim=True).values
    e = torch.exp(s)
    a = e / e.sum(dim=-1, keepdim=True)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return linear(o, p[prefix + "proj.weight"], p[prefix + "proj.bias"])


def mlp(x: torch.Tensor, p: Dict[str, torch.Tensor], prefix: str = "mlp.") -> torch.Tensor:
    """Mlp.forward, PointCloud/openpoints/models/layers/mlp.py:29-35: fc1 -> GELU -> fc2
    (dropouts are identity in eval / p=0)."""
    h = linear(x, p[prefix + "fc1.weight"], p[prefix + "fc1.bias"])
    return linear(gelu_erf(h), p[prefix + "fc2.weight"], p[prefix + "fc2.bias"])


def windowed_attention(x: torch.Tensor, p: Dict[str, torch.Tensor], num_heads: int, H: int, W: int, window: int) -> torch.Tensor:
    """WindowedAttention.forward, Image/detection/mmdet_custom/models/backbones/base/vit.py:160-190, restated with plain
    indexing instead of F.unfold / F.fold: qkv Linear on every token (:166); the [B, H, W, 3C] grid is zero-padded on the
    bottom/right to multiples of the window (:167-168) -- so padded positions hold q = k = v = 0 and DO take part in the
    softmax as keys with score 0; windows are (wy, wx) row-major with (iy, ix) row-major inside (:170-175); attention per
    window and head (:179-183); fold back, crop to H x W (:185-189); proj (:190)."""
    B, N, C = x.shape
    hd = C // num_heads
    scale = hd ** -0.5
    gh, gw = -(-H // window), -(-W // window)
    qkv = x @ p["attn.qkv.weight"].t()
    if "attn.qkv.bias" in p and p["attn.qkv.bias"] is not None:
        qkv = qkv + p["attn.qkv.bias"]
    grid = qkv.new_zeros(B, gh * window, gw * window, 3 * C)
    grid[:, :H, :W] = qkv.reshape(B, H, W, 3 * C)
    wins = grid.reshape(B, gh, window, gw, window, 3, num_heads, hd).permute(5, 0, 1, 3, 6, 2, 4, 7)
    wins = wins.reshape(3, B, gh * gw, num_heads, window * window, hd)
    q, k, v = wins[0], wins[1], wins[2]
    attn = torch.softmax((q @ k.transpose(-2, -1)) * scale, dim=-1)
    o = attn @ v                                                            # [B, L, heads, ws*ws, hd]
    o = o.reshape(B, gh, gw, num_heads, window, window, hd).permute(0, 1, 4, 2, 5, 3, 6)
    o = o.reshape(B, gh * window, gw * window, C)[:, :H, :W].reshape(B, N, C)
    return o @ p["attn.proj.weight"].t() + p["attn.proj.bias"]


def block_forward(x: torch.Tensor, p: Dict[str, torch.Tensor], num_heads: int, eps: float = 1e-5,
                  gamma1: Optional[torch.Tensor] = None, gamma2: Optional[torch.Tensor] = None,
                  pre_scale_q: bool = False, window: Optional[tuple] = None) -> torch.Tensor:
    """Block.forward, PointCloud/openpoints/models/layers/attention.py:55-58:
        x = x + drop_path(attn(norm1(x)));  x = x + drop_path(mlp(norm2(x)))
    gamma1/gamma2 restate the layer-scale variant
    (Image/detection/mmdet_custom/models/backbones/base/vit.py:313-316)."""
    if window is not None:      # (H, W, window_size): the windowed blocks of the detection backbone (vit.py:284-287)
        a = windowed_attention(layer_norm(x, p["norm1.weight"], p["norm1.bias"], eps), p, num_heads, *window)
    else:
        a = attention(layer_norm(x, p["norm1.weight"], p["norm1.bias"], eps), p, num_heads,
                      pre_scale_q=pre_scale_q)
    x = x + (a if gamma1 is None else gamma1 * a)
    m = mlp(layer_norm(x, p["norm2.weight"], p["norm2.bias"], eps), p)
    x = x + (m if gamma2 is None else gamma2 * m)
    return x


def video_block_params(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Video/models/modeling_finetune.py key set -> the names block_forward() reads.  The Video Attention has a bias-free
    qkv Linear plus q_bias / v_bias and builds `cat(q_bias, zeros_like(v_bias), v_bias)` at run time (:172-178): the K
    third of the bias is identically zero.  gamma_1 / gamma_2 (:245-259) are the layer-scale vectors."""
    p = {k: v for k, v in sd.items() if k not in ("attn.q_bias", "attn.v_bias", "gamma_1", "gamma_2")}
    p["attn.qkv.bias"] = torch.cat([sd["attn.q_bias"], torch.zeros_like(sd["attn.v_bias"]), sd["attn.v_bias"]])
    p["gamma1"], p["gamma2"] = sd["gamma_1"], sd["gamma_2"]
    return p


def split_state_dict(sd: Dict[str, torch.Tensor]) -> "list[Dict[str, torch.Tensor]]":
    """'{i}.name' -> per-block dicts, in block order (nn.Sequential key layout, README.md:125-135)."""
    blocks: Dict[int, Dict[str, torch.Tensor]] = {}
    for k, v in sd.items():
        i, name = k.split(".", 1)
        blocks.setdefault(int(i), {})[name] = v
    return [blocks[i] for i in sorted(blocks)]


def encoder_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], num_heads: int, eps: float = 1e-5,
                    pos_embed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Sequential(*[Block]*L)(x)  (README.md:124-149).  With ``pos_embed`` the PointCloud
    per-block re-injection is restated: ``for block in blocks: x = block(x + pos_embed)``
    (PointCloud/openpoints/models/backbone/metatransformer.py:161-163)."""
    x = x.float()
    for p in split_state_dict(sd):
        if pos_embed is not None:
            x = x + pos_embed
        x = block_forward(x, p, num_heads, eps)
    return x


def encoder_forward_backward(x: torch.Tensor, sd: Dict[str, torch.Tensor], num_heads: int,
                             grad_out: torch.Tensor, eps: float = 1e-5):
    """Forward + autograd backward of the restated encoder: returns (y, dx, {name: dparam}).
    Loss = sum(y * grad_out), i.e. grad_out is dL/dy."""
    xs = x.detach().float().clone().requires_grad_(True)
    params = OrderedDict((k, v.detach().float().clone().requires_grad_(True)) for k, v in sd.items())
    y = encoder_forward(xs, params, num_heads, eps)
    (y * grad_out).sum().backward()
    return y.detach(), xs.grad.detach(), OrderedDict((k, v.grad.detach()) for k, v in params.items())
