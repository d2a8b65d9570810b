"""GPU: task heads and encoder checkpoint I/O -- token pooling, the Video classification tail against
fixtures generated from the reference's own VisionTransformer modules, the PointCloud ClsHead against its torch restatement,
and the .pth wire format <-> packed device layout round trip."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import GOLDEN, TOL_BF16_OP, TOL_F32, check_close, rel_err
import metatransformer_amd as M
from metatransformer_amd import heads
from oracle import block_oracle as bo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["mean", "max", "cls"])
def test_pool_tokens_forward_backward(dev, mode, dt):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(5, 37, 96, generator=g).to(dt)
    xr = x.float().clone().requires_grad_(True)
    ref = {"mean": lambda t: t.mean(1), "max": lambda t: t.max(1)[0], "cls": lambda t: t[:, 0]}[mode](xr)
    go = torch.randn(5, 96, generator=g)
    (ref * go).sum().backward()
    xd = x.detach().clone().to(dev).requires_grad_(True)
    y = heads.pool_tokens(xd, mode)
    assert y.dtype == torch.float32 and y.shape == (5, 96)
    (y * go.to(dev)).sum().backward()
    if mode in ("max", "cls"):                                   # selections: exact
        assert torch.equal(y.cpu(), ref.detach())
    else:
        assert rel_err(y, ref) < 1e-6
    assert rel_err(xd.grad.float(), xr.grad) < (1e-6 if dt == torch.float32 else 4e-3)      # (bf16: dx is stored in bf16)


@pytest.mark.parametrize("tag", ["head_mean", "head_cls"])
def test_video_classifier_head_matches_reference_golden(dev, tag):
    """fc_norm(x.mean(1)) -> head and norm(x)[:, 0] -> head (Video/models/modeling_finetune.py:445-460); 174 classes."""
    z = np.load(os.path.join(GOLDEN, "variants.npz"))
    T = lambda k: torch.from_numpy(z[f"{tag}/{k}"])              # noqa: E731
    h = heads.ClassifierHead(64, 174, use_mean_pooling=tag == "head_mean")
    ln = "fc_norm" if tag == "head_mean" else "norm"
    assert set(h.state_dict()) == {f"{ln}.weight", f"{ln}.bias", "head.weight", "head.bias"}        # the reference's names
    h.load_state_dict({f"{ln}.weight": T("ln_w"), f"{ln}.bias": T("ln_b"), "head.weight": T("head_w"), "head.bias": T("head_b")},
                      strict=True)
    h = h.to(dev).eval()
    x = T("x").to(dev).requires_grad_(True)
    y = h(x)
    (y * T("go").to(dev)).sum().backward()
    check_close(y, T("y"), TOL_F32, tag + " logits")
    check_close(x.grad, T("dx"), TOL_F32, tag + " dx")
    lnm = getattr(h, ln)
    for got, want in ((h.head.weight.grad, "dhead_w"), (h.head.bias.grad, "dhead_b"), (lnm.weight.grad, "dln_w"), (lnm.bias.grad, "dln_b")):
        assert rel_err(got, T(want)) < TOL_F32, want


def test_pointcloud_cls_head_vs_restatement(dev):
    """ClsHead(global_feat='max,avg', mlps=[256, 256], bn1d, relu, dropout) in eval mode vs the same modules in plain torch
    (openpoints/models/classification/cls_base.py:77-136; channel-first [B, C, N] input, point_dim=2, as the reference)."""
    torch.manual_seed(3)
    h = heads.ClsHead(num_classes=40, in_channels=768, mlps=[256, 256], norm_args={"norm": "bn1d"}, global_feat="max,avg", point_dim=2)
    keys = list(h.state_dict())
    assert keys[0] == "head.0.0.weight" and "head.0.1.running_mean" in keys and keys[-1] == "head.4.0.bias"
    for m in h.modules():
        if isinstance(m, nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.normal_(1, 0.1); m.bias.data.normal_(0, 0.1)
    h.eval()
    f = torch.randn(4, 768, 129)                                 # [B, C, N]
    ref = torch.cat([f.max(dim=2)[0], f.mean(dim=2)], dim=1)
    for m in h.head:
        ref = m(ref)
    hd = h.to(dev)
    fd = f.to(dev).requires_grad_(True)
    y = hd(fd)
    assert rel_err(y, ref.detach()) < TOL_F32
    y.square().sum().backward()
    assert fd.grad is not None and torch.isfinite(fd.grad).all() and fd.grad.abs().sum() > 0


def test_checkpoint_io_roundtrip_and_packed_layout(dev, tmp_path):
    sd = bo.make_encoder_state_dict(2, 128, seed=21)
    x = torch.randn(2, 33, 128, generator=torch.Generator().manual_seed(1))
    y_ref = bo.encoder_forward(x, sd, 2)
    # (1) the reference wire format, written by torch.save as the released .pth files are, under wrapper prefixes, in fp16
    path = tmp_path / "enc.pth"
    torch.save({"model": {"module.blocks." + k: v.half() for k, v in sd.items()} | {"module.head.weight": torch.zeros(3, 3)}}, path)
    enc = M.build_encoder(2, 128, 2)
    heads.load_encoder_checkpoint(enc, str(path), strict=True)
    assert all(p.dtype == torch.float32 for p in enc.parameters())
    assert rel_err(enc.state_dict()["1.mlp.fc2.weight"], sd["1.mlp.fc2.weight"].half().float()) == 0.0
    # (2) exact fp32 round trip through save_encoder_checkpoint
    enc.load_state_dict(sd, strict=True)
    out = tmp_path / "out.pth"
    heads.save_encoder_checkpoint(enc, out)
    back = torch.load(out, weights_only=True)
    assert list(back) == list(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    # (3) packed device layout: parameters become views of one flat fp32 buffer, no-decay tensors first, compute copies warm
    flat = heads.pack_encoder(enc, device=dev)
    assert 0 < flat.no_decay_numel < flat.numel
    assert all(p.data_ptr() == flat.flat_param.data_ptr() + 4 * o for p, o in zip(flat.params, flat.offsets))
    with torch.no_grad():
        y = enc(x.to(dev))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y16 = enc(x.to(dev))
    assert rel_err(y, y_ref) < TOL_F32 and rel_err(y16, y_ref) < 1e-2
    # (4) the Video key set loads through the same entry point
    vsd = {}
    for k, v in sd.items():
        if k.endswith("attn.qkv.bias"):
            vsd["blocks." + k[:-8] + "q_bias"], vsd["blocks." + k[:-8] + "v_bias"] = v[:128].clone(), v[256:].clone()
        else:
            vsd["blocks." + k] = v
    for i in range(2):
        vsd[f"blocks.{i}.gamma_1"], vsd[f"blocks.{i}.gamma_2"] = torch.full((128,), 0.5), torch.full((128,), 0.25)
    venc = M.build_encoder(2, 128, 2, layer_scale=True)
    heads.load_encoder_checkpoint(venc, vsd, strict=True)
    assert torch.equal(venc[1].gamma2.data, torch.full((128,), 0.25)) and torch.equal(venc[0].attn.qkv.bias.data[128:256], torch.zeros(128))


@pytest.mark.parametrize("n,m", [(1024, 256), (777, 100), (64, 64), (2048, 128)])
def test_fps_indices_bit_exact(dev, n, m):
    """farthest point sampling: the index sequence of the reference kernel's rule, bit for bit -- also with duplicated
    points (ties), where the thread-strided scan + tree fold decide"""
    from oracle import tokenizer_oracle as to
    g = torch.Generator().manual_seed(n + m)
    p = torch.rand(3, n, 3, generator=g) * 2 - 1
    p[1, n // 2:] = p[1, : n - n // 2].clone()                   # exact duplicates: ties
    p[2, :] = torch.round(p[2] * 4) / 4                          # a coarse lattice: many equal distances
    idx = heads.furthest_point_sample(p.to(dev), m).cpu().numpy()
    ref = to.fps_reference(p.numpy(), m)
    assert idx.dtype == np.int32 and (idx[:, 0] == 0).all()
    assert np.array_equal(idx, ref)


def test_knn_indices_match_cdist_topk(dev):
    from oracle import tokenizer_oracle as to
    g = torch.Generator().manual_seed(9)
    p = torch.rand(2, 1024, 3, generator=g)
    c = p[:, ::4].contiguous()
    got = heads.knn_indices(p.to(dev), c.to(dev), 32).cpu().long()
    ref = to.knn_reference(p, c, 32).long()
    assert got.shape == ref.shape == (2, 256, 32)
    assert torch.equal(got[:, :, 0], torch.arange(0, 1024, 4).expand(2, -1))          # a centre's nearest point is itself
    # same neighbour SETS everywhere; the ORDER is nearest-first in exact arithmetic (cdist goes through a matmul, ~1e-6
    # absolute on squared distances, so its own order of near-ties is not exact): distances non-decreasing, and equal to
    # the reference's k smallest distances position by position
    assert torch.equal(got.sort(dim=2).values, ref.sort(dim=2).values)
    d_all = (c.double().unsqueeze(2) - p.double().unsqueeze(1)).pow(2).sum(-1).sqrt()       # [B, m, n] exact
    d_got, d_ref = d_all.gather(2, got), d_all.gather(2, ref)
    assert bool((d_got[:, :, 1:] >= d_got[:, :, :-1] - 1e-7).all())
    assert (d_got - d_ref).abs().max() < 1e-4


def test_point_patch_embed_vs_restatement(dev):
    """tokens of a ModelNet-sized cloud ([B, 1024, 3] -> [B, 256, 768]) against PointPatchEmbed.forward restated on the same
    module in plain torch (group_embed.py:138-172), eval mode; indices first (bit-exact), then the features"""
    from oracle import tokenizer_oracle as to
    torch.manual_seed(11)
    mod = heads.PointPatchEmbed(sample_ratio=0.25, group_size=32, embed_dim=768, channels=(128, 256, 512))
    assert list(mod.state_dict())[:3] == ["conv1.0.0.weight", "conv1.0.1.weight", "conv1.0.1.bias"]
    assert mod.conv1[0][0].weight.shape == (128, 3, 1, 1) and mod.conv2[0][0].weight.shape == (512, 512, 1, 1)
    for m in mod.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    mod.eval()
    p = torch.rand(2, 1024, 3, generator=torch.Generator().manual_seed(12)) * 2 - 1
    ref, ridx, rnbr = to.point_patch_embed_reference(p, mod)
    md = mod.to(dev)
    out, center, idx, nbr = md.tokens(p.to(dev))
    assert torch.equal(idx.cpu().long(), ridx)
    assert torch.equal(nbr.cpu().long().sort(dim=2).values, rnbr.sort(dim=2).values)
    assert out.shape == (2, 256, 768)
    assert rel_err(out.transpose(1, 2), ref.detach()) < TOL_F32
    (pc, cc), (_, of) = md(p.to(dev))                            # the reference's return convention
    assert of.shape == (2, 768, 256) and cc.shape == (2, 256, 3)
