"""Generate tests/golden/*.npz by running the REFERENCE'S OWN code (imported unmodified from
/root/reference through oracle/ref_loader.py) on seeded inputs.  TEST INFRASTRUCTURE.

    python -m oracle.make_golden            # (re)write the fixtures
    python -m oracle.make_golden --check    # also check oracle/block_oracle.py against the reference

Runs only in the build container (needs /root/reference).  The fixtures travel to the GPU
box; the reference does not.  Weights are regenerated on both sides from
oracle.block_oracle.make_encoder_state_dict(seed) -- each file stores the generator's checksum so a
drifting RNG is caught -- except the 'tiny' case, which stores its weights in full.

Large outputs are stored token-subsampled (``tok_stride``) to keep the fixtures small.
"""
from __future__ import annotations

import argparse
import json
import os

import numpy as np
import torch

from . import block_oracle as bo
from . import ref_loader
from . import tokenizer_oracle as to

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# name: depth, dim, heads, B, N, eps, seed, tok_stride, with_backward
ENCODER_CASES = {
    "tiny":        dict(depth=2, dim=32, heads=2, B=2, N=5, eps=1e-5, seed=11, tok_stride=1, backward=True, store_weights=True),
    "small_hd64":  dict(depth=2, dim=128, heads=2, B=3, N=37, eps=1e-5, seed=12, tok_stride=1, backward=True, grad_stride=1),
    "base_1blk":   dict(depth=1, dim=768, heads=12, B=2, N=197, eps=1e-5, seed=13, tok_stride=8, backward=True, grad_stride=5),
    "base_12blk":  dict(depth=12, dim=768, heads=12, B=1, N=197, eps=1e-5, seed=14, tok_stride=8, backward=False),
    "base_eps1e6": dict(depth=1, dim=768, heads=12, B=1, N=50, eps=1e-6, seed=15, tok_stride=2, backward=False),
    "large_2blk":  dict(depth=2, dim=1024, heads=16, B=1, N=64, eps=1e-5, seed=16, tok_stride=4, backward=False),
    "graph_hd24":  dict(depth=1, dim=768, heads=32, B=2, N=40, eps=1e-5, seed=17, tok_stride=2, backward=False),
}


def _inputs(c):
    g = torch.Generator().manual_seed(1000 + c["seed"])
    x = torch.randn(c["B"], c["N"], c["dim"], generator=g)
    go = torch.randn(c["B"], c["N"], c["dim"], generator=g)
    return x, go


def gen_encoder_case(name, c, check):
    sd = bo.make_encoder_state_dict(c["depth"], c["dim"], seed=c["seed"])
    enc = ref_loader.reference_encoder(c["depth"], c["dim"], c["heads"], eps=c["eps"])
    enc.load_state_dict(sd, strict=True)           # strict=True as every call site does
    x, go = _inputs(c)
    out = {"config": json.dumps({k: v for k, v in c.items()}), "weights_checksum": np.float64(bo.state_dict_checksum(sd))}
    s = c["tok_stride"]
    if c["backward"]:
        xr = x.clone().requires_grad_(True)
        y = enc(xr)
        (y * go).sum().backward()
        out["dx"] = xr.grad[:, ::s].numpy()
        # parameter grads: store per-tensor (sum, abs-sum, first 16 flat values) -- compact but discriminating
        for k, p in enc.named_parameters():
            g = p.grad.double()
            out["dparam_stats/" + k] = np.array([g.sum().item(), g.abs().sum().item()])
            out["dparam_head/" + k] = p.grad.flatten()[:16].numpy()
            # grad_stride: every gs-th element of the flattened gradient, the reference's own values (the statistics
            # above pin a gradient only through the restatement).  gs = 1 keeps it whole; gs = 5 is coprime to every row length
            # of the Base shapes (768, 2304, 3072), so every row AND every column of every weight gradient is sampled and the
            # Base block stays at 5.7 MB instead of 28
            gs = c.get("grad_stride", 0)
            if gs:
                out["dparam_s/" + k] = p.grad.flatten()[::gs].numpy().copy()
        y = y.detach()
        for p in enc.parameters():
            p.grad = None
    else:
        with torch.no_grad():
            y = enc(x)
    out["y"] = y[:, ::s].numpy()
    if c["N"] * c["B"] * c["dim"] <= 64 * 1024:
        out["x"] = x.numpy()
        out["grad_out"] = go.numpy()
    if c.get("store_weights"):
        for k, v in sd.items():
            out["w/" + k] = v.numpy()
    if check:
        yo = bo.encoder_forward(x, sd, c["heads"], c["eps"])
        err = (yo - y).abs().max().item() / y.abs().max().item()
        assert err < 2e-6, (name, err)
        print(f"  oracle-vs-reference {name}: rel {err:.2e}")
    np.savez(os.path.join(GOLDEN_DIR, f"encoder_{name}.npz"), **out)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in out.items() if k in ("y", "dx")})


def gen_tokenizers(check):
    out = {}
    # Image: Data2Seq/Image.py PatchEmbed, 224x224 patch16 (BASELINE config 1)
    g = torch.Generator().manual_seed(2001)
    PE = ref_loader.reference_image_patch_embed()
    pe = PE(img_size=224, patch_size=16, in_c=3, embed_dim=768).eval()
    w = 0.02 * torch.randn(pe.proj.weight.shape, generator=g)
    b = 0.05 * torch.randn(pe.proj.bias.shape, generator=g)
    pe.proj.weight.data.copy_(w); pe.proj.bias.data.copy_(b)
    x = torch.randn(2, 3, 224, 224, generator=g)
    with torch.no_grad():
        y = pe(x)
    out["image/y"] = y[:, ::7, ::3].numpy()
    out["image/seed"] = np.int64(2001)
    if check:
        yo = to.image_patch_embed(x, w, b)
        print("  image tokenizer oracle rel", ((yo - y).abs().max() / y.abs().max()).item())
        assert torch.allclose(yo, y, atol=2e-5)

    # Acoustic: Data2Seq/Acoustic.py PatchEmbed k16 stride 10, spectrogram 128 x 100
    g = torch.Generator().manual_seed(2002)
    # Data2Seq/Acoustic.py:16 cannot be instantiated as written (it wraps an already-2-tupled
    # patch_size in another tuple -> nn.Conv2d raises); the working construction is the AST one,
    # Audio/src/models/ast_models.py:86: nn.Conv2d(1, C, kernel_size=(16,16), stride=(fstride,tstride)),
    # followed by .flatten(2).transpose(1,2) (Data2Seq/Acoustic.py:22 / ast_models.py:31).
    proj = torch.nn.Conv2d(1, 768, kernel_size=(16, 16), stride=(10, 10)).eval()
    w = 0.02 * torch.randn(proj.weight.shape, generator=g)
    b = 0.05 * torch.randn(proj.bias.shape, generator=g)
    proj.weight.data.copy_(w); proj.bias.data.copy_(b)
    x = torch.randn(2, 1, 128, 100, generator=g)
    with torch.no_grad():
        y = proj(x).flatten(2).transpose(1, 2)
    out["acoustic/y"] = y[:, ::3, ::3].numpy()
    out["acoustic/tokens"] = np.int64(y.shape[1])
    if check:
        yo = to.acoustic_patch_embed(x, w, b)
        assert yo.shape == y.shape and torch.allclose(yo, y, atol=2e-5)
        print("  acoustic tokenizer oracle OK", tuple(y.shape))

    # Video tubelet: Video/models/modeling_finetune.py PatchEmbed, small clip 3x4x64x64, tubelet 2
    g = torch.Generator().manual_seed(2003)
    V = ref_loader.reference_video_module()
    vp = V.PatchEmbed(img_size=64, patch_size=16, in_chans=3, embed_dim=768, num_frames=4, tubelet_size=2).eval()
    w = 0.02 * torch.randn(vp.proj.weight.shape, generator=g)
    b = 0.05 * torch.randn(vp.proj.bias.shape, generator=g)
    vp.proj.weight.data.copy_(w); vp.proj.bias.data.copy_(b)
    x = torch.randn(2, 3, 4, 64, 64, generator=g)
    with torch.no_grad():
        y = vp(x)
    out["video/y"] = y[:, :, ::3].numpy()
    tab = V.get_sinusoid_encoding_table(32, 768)[0]
    out["video/sinusoid_head"] = tab[:, :16].numpy()
    if check:
        yo = to.video_tubelet_embed(x, w, b)
        assert torch.allclose(yo, y, atol=2e-5)
        assert torch.allclose(to.sinusoid_table_video(32, 768), tab, atol=1e-6)
        print("  video tokenizer oracle OK", tuple(y.shape))

    # Time series: Data2Seq/Time_Series.py DataEmbedding(c_in=7, d_model=768, 'fixed', freq 'h')
    g = torch.Generator().manual_seed(2004)
    TS = ref_loader.reference_time_series_embedding()
    torch.manual_seed(2004)                                    # the reference ctor draws its kaiming init from the global RNG
    ts = TS(c_in=7, d_model=768, embed_type="fixed", freq="h", dropout=0.1).eval()
    w = ts.value_embedding.tokenConv.weight.data.clone()       # kaiming init from the reference ctor
    out["ts/conv_weight"] = w.numpy()
    x = torch.randn(2, 96, 7, generator=g)
    mark = torch.stack([torch.randint(0, 13, (2, 96), generator=g), torch.randint(0, 32, (2, 96), generator=g),
                        torch.randint(0, 7, (2, 96), generator=g), torch.randint(0, 24, (2, 96), generator=g)], dim=-1).float()
    with torch.no_grad():
        y = ts(x, mark)
        y_nomark = ts(x, None)
    out["ts/x"] = x.numpy(); out["ts/mark"] = mark.numpy()
    out["ts/y"] = y[:, ::4, ::3].numpy()
    out["ts/y_nomark"] = y_nomark[:, ::4, ::3].numpy()
    if check:
        tabs = [to.sinusoid_table_ts(n, 768) for n in (13, 32, 7, 24)]
        yo = to.time_series_embedding(x, w, mark, tabs, to.sinusoid_table_ts(5000, 768))
        err = (yo - y).abs().max().item()
        assert err < 2e-5, err
        yo2 = to.time_series_embedding(x, w, None, None, to.sinusoid_table_ts(5000, 768))
        assert (yo2 - y_nomark).abs().max().item() < 2e-5
        print("  time-series tokenizer oracle OK max-abs", err)
    np.savez(os.path.join(GOLDEN_DIR, "tokenizers.npz"), **out)
    print("wrote tokenizers")


def _randomize(module, g, skip=()):
    """seeded N(0, .) values for every parameter, so that the fixture is sensitive to each of them"""
    for k, p in module.named_parameters():
        if k in skip:
            continue
        if k.startswith("gamma"):
            p.data.copy_(0.5 + 0.5 * torch.rand(p.shape, generator=g))
        elif p.dim() == 1 and "norm" in k and k.endswith("weight"):
            p.data.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
        elif p.dim() == 1:
            p.data.copy_(0.05 * torch.randn(p.shape, generator=g))
        else:
            p.data.copy_(0.05 * torch.randn(p.shape, generator=g))


def _run_block(blk, x, go, *fwd_args):
    xr = x.clone().requires_grad_(True)
    y = blk(xr, *fwd_args)
    (y * go).sum().backward()
    grads = {k: p.grad.clone() for k, p in blk.named_parameters()}
    for p in blk.parameters():
        p.grad = None
    return y.detach(), xr.grad.detach(), grads


def _store_grads(out, tag, grads):
    """parameter gradients: (sum, abs-sum) + the first 256 flat values per tensor (1-D tensors in full)"""
    for k, v in grads.items():
        out[f"{tag}/dw_stats/" + k] = np.array([v.double().sum().item(), v.double().abs().sum().item()])
        out[f"{tag}/dw_head/" + k] = v.flatten()[:256].numpy() if v.dim() > 1 else v.numpy()


def gen_variants(check):
    """Block variants of the big pipelines and the pos-embed table resize (a16), from the reference's
    own classes: Video/models/modeling_finetune.py Block (q/v-only bias :160-166, q pre-scaled :183-184, gamma_1/gamma_2
    :245-259) and Image/detection/.../base/vit.py Block (WindowedAttention :148-192, gamma1/gamma2 :298-320,
    resize_pos_embed :459-486)."""
    from functools import partial
    import torch.nn as nn
    out = {}
    # ---- Video block: dim 128, 2 heads (hd 64), eps 1e-6 as vit_base_patch16_224 builds it
    VM = ref_loader.reference_video_module()
    g = torch.Generator().manual_seed(3001)
    vb = VM.Block(128, 2, qkv_bias=True, init_values=0.1, norm_layer=partial(nn.LayerNorm, eps=1e-6)).eval()
    _randomize(vb, g)
    x, go = torch.randn(2, 40, 128, generator=g), torch.randn(2, 40, 128, generator=g)
    y, dx, grads = _run_block(vb, x, go)
    out.update({"video/x": x.numpy(), "video/go": go.numpy(), "video/y": y.numpy(), "video/dx": dx.numpy()})
    for k, v in vb.state_dict().items():
        out["video/w/" + k] = v.numpy()
    _store_grads(out, "video", grads)
    if check:
        p = bo.video_block_params(vb.state_dict())
        yo = bo.block_forward(x, p, 2, 1e-6, gamma1=p["gamma1"], gamma2=p["gamma2"], pre_scale_q=True)
        err = ((yo - y).abs().max() / y.abs().max()).item()
        assert err < 2e-6, err
        print("  video block oracle rel", err)

    # ---- detection blocks: windowed + layer-scale on a ragged 9 x 11 grid with 4 x 4 windows; full attention + layer-scale
    DV = ref_loader.reference_detection_vit_module()
    for tag, kw, hw in (("det_win", dict(windowed=True, window_size=4, layer_scale=True), (9, 11)),
                        ("det_full", dict(windowed=False, layer_scale=True), (5, 7))):
        g = torch.Generator().manual_seed(3002 if tag == "det_win" else 3003)
        db = DV.Block(128, 2, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kw).eval()
        _randomize(db, g)
        H, W = hw
        x, go = torch.randn(2, H * W, 128, generator=g), torch.randn(2, H * W, 128, generator=g)
        y, dx, grads = _run_block(db, x, go, H, W)
        out.update({f"{tag}/x": x.numpy(), f"{tag}/go": go.numpy(), f"{tag}/y": y.numpy(), f"{tag}/dx": dx.numpy(),
                    f"{tag}/hw": np.array([H, W, kw.get("window_size", 0)])})
        for k, v in db.state_dict().items():
            out[f"{tag}/w/" + k] = v.numpy()
        _store_grads(out, tag, grads)
        if check:
            p = dict(db.state_dict())
            yo = bo.block_forward(x, p, 2, 1e-6, gamma1=p["gamma1"], gamma2=p["gamma2"],
                                  window=(H, W, kw["window_size"]) if kw["windowed"] else None)
            err = ((yo - y).abs().max() / y.abs().max()).item()
            assert err < 2e-6, (tag, err)
            print(f"  {tag} oracle rel", err)
    # Base-sized windowed block: 768-d, 12 heads, 14 x 14 windows on a 20 x 30 grid (forward only, subsampled)
    g = torch.Generator().manual_seed(3004)
    db = DV.Block(768, 12, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), windowed=True, window_size=14, layer_scale=True).eval()
    sd = bo.make_encoder_state_dict(1, 768, seed=3004)
    db.load_state_dict({**{k[2:]: v for k, v in sd.items()}, "gamma1": db.gamma1.data, "gamma2": db.gamma2.data}, strict=True)
    db.gamma1.data.copy_(0.5 + 0.5 * torch.rand(768, generator=g)); db.gamma2.data.copy_(0.5 + 0.5 * torch.rand(768, generator=g))
    x = torch.randn(1, 600, 768, generator=g)
    with torch.no_grad():
        y = db(x, 20, 30)
    out.update({"det_win_base/y": y[:, ::5, ::3].numpy(), "det_win_base/gamma1": db.gamma1.data.numpy(),
                "det_win_base/gamma2": db.gamma2.data.numpy(), "det_win_base/weights_checksum": np.float64(bo.state_dict_checksum(sd))})

    # ---- pos-embed table resize (bicubic, align_corners=False), cls row kept: 14 x 14 -> 20 x 24 and -> 10 x 7
    g = torch.Generator().manual_seed(3005)
    pos = torch.randn(1, 1 + 14 * 14, 96, generator=g)
    out["resize/pos"] = pos.numpy()
    for tag, shp in (("up", (20, 24)), ("down", (10, 7))):
        out[f"resize/{tag}"] = DV.TIMMVisionTransformer.resize_pos_embed(pos, shp, (14, 14), "bicubic").numpy()
    out["resize/bilinear_up"] = DV.TIMMVisionTransformer.resize_pos_embed(pos, (20, 24), (14, 14), "bilinear").numpy()
    # ---- classification tail of the Video model (modeling_finetune.py:395, 445-454): fc_norm(x.mean(1)) -> head, and the
    # cls-token form norm(x)[:, 0] -> head, on the reference's own modules (174 classes: not a multiple of 8)
    g = torch.Generator().manual_seed(3006)
    for tag, mean_pool in (("head_mean", True), ("head_cls", False)):
        vt = VM.VisionTransformer(img_size=32, patch_size=16, in_chans=3, num_classes=174, embed_dim=64, depth=1, num_heads=2,
                                  mlp_ratio=4, qkv_bias=True, init_values=0.0, all_frames=4, tubelet_size=2,
                                  use_mean_pooling=mean_pool, init_scale=1.0).eval()
        ln = vt.fc_norm if mean_pool else vt.norm
        ln.weight.data.copy_(1.0 + 0.1 * torch.randn(64, generator=g)); ln.bias.data.copy_(0.05 * torch.randn(64, generator=g))
        vt.head.weight.data.copy_(0.1 * torch.randn(174, 64, generator=g)); vt.head.bias.data.copy_(0.05 * torch.randn(174, generator=g))
        x = torch.randn(3, 9, 64, generator=g, requires_grad=True)
        go = torch.randn(3, 174, generator=g)
        feat = vt.fc_norm(x.mean(1)) if mean_pool else vt.norm(x)[:, 0]        # the tail of forward_features (:445-454)
        y = vt.head(vt.head_dropout(feat))                                     # forward (:456-460)
        (y * go).sum().backward()
        out.update({f"{tag}/x": x.detach().numpy(), f"{tag}/go": go.numpy(), f"{tag}/y": y.detach().numpy(), f"{tag}/dx": x.grad.numpy(),
                    f"{tag}/ln_w": ln.weight.data.numpy(), f"{tag}/ln_b": ln.bias.data.numpy(), f"{tag}/head_w": vt.head.weight.data.numpy(),
                    f"{tag}/head_b": vt.head.bias.data.numpy(), f"{tag}/dhead_w": vt.head.weight.grad.numpy(),
                    f"{tag}/dhead_b": vt.head.bias.grad.numpy(), f"{tag}/dln_w": ln.weight.grad.numpy(), f"{tag}/dln_b": ln.bias.grad.numpy()})
    np.savez(os.path.join(GOLDEN_DIR, "variants.npz"), **out)
    print("wrote variants", sum(v.nbytes for v in out.values()) // 1024, "KiB")


def gen_pointcloud(check):
    """Row f4 (point-cloud front end and head) from the reference's own classes: PointPatchEmbed
    (PointCloud/openpoints/models/layers/group_embed.py:60-172) in the configuration of
    cfgs/modelnet40ply2048/metatransformer.yaml:19-33 (fps / knn / 'dp' / bn / conv-norm-act / max; narrower channels to
    keep the fixture small) and ClsHead (models/classification/cls_base.py:77-136) in three constructions.  The two CUDA
    extension calls inside are served as ref_loader.reference_pointcloud_modules documents."""
    ge, cb = ref_loader.reference_pointcloud_modules()
    out = {}
    g = torch.Generator().manual_seed(4001)

    def randomize(mod):
        for m in mod.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
                m.weight.data.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.data.copy_(0.05 * torch.randn(m.bias.shape, generator=g))
            elif isinstance(m, (torch.nn.Linear, torch.nn.Conv2d)):
                m.weight.data.copy_(torch.randn(m.weight.shape, generator=g) * (m.weight[0].numel() ** -0.5))
                if m.bias is not None:
                    m.bias.data.copy_(0.05 * torch.randn(m.bias.shape, generator=g))

    # ---- PointPatchEmbed: [2, 512, 3] -> centres [2, 128, 3], tokens [2, 192, 128]
    cfg = dict(sample_ratio=0.25, group_size=32, in_channels=3, layers=4, embed_dim=192, channels=[32, 64, 128], subsample="fps",
               group="knn", feature_type="dp", normalize_dp=False, norm_args={"norm": "bn"}, conv_args={"order": "conv-norm-act"},
               reduction="max")
    ppe = ge.PointPatchEmbed(**cfg).eval()
    randomize(ppe)
    p = torch.rand(2, 512, 3, generator=g) * 2 - 1
    p[1, 300:] = p[1, :212].clone()                       # duplicated points: ties in both the sampler and the neighbour search
    with torch.no_grad():
        (_, center), (_, out_f) = ppe(p)
    out.update({"ppe/config": json.dumps(cfg), "ppe/p": p.numpy(), "ppe/center": center.numpy(), "ppe/out_f": out_f.numpy()})
    for k, v in ppe.state_dict().items():
        out["ppe/w/" + k] = v.numpy()
    if check:
        ref, _, _ = to.point_patch_embed_reference(p, ppe)
        err = ((ref - out_f).abs().max() / out_f.abs().max()).item()
        assert err < 1e-5, err
        print("  point_patch_embed restatement rel", err)

    # ---- ClsHead: (a) the pipeline's construction (bn1d, two hidden layers), eval; (b) the class defaults (no norm, in-place
    # ReLU straight behind the Linear), eval + a training-mode forward/backward with dropout off; (c) global_feat 'max,avg'
    # over channel-first features (point_dim = 2, the default)
    cases = {"a": (dict(num_classes=40, in_channels=192, mlps=[64, 64], norm_args={"norm": "bn1d"}), (5, 192)),
             "b": (dict(num_classes=10, in_channels=64, dropout=0.0), (6, 64)),
             "c": (dict(num_classes=12, in_channels=48, mlps=[32], global_feat="max,avg"), (3, 48, 17))}
    for tag, (kw, shape) in cases.items():
        head = cb.ClsHead(**kw).eval()
        randomize(head)
        x = torch.randn(*shape, generator=g)
        with torch.no_grad():
            y = head(x)
        out.update({f"cls_{tag}/config": json.dumps(kw), f"cls_{tag}/x": x.numpy(), f"cls_{tag}/y": y.numpy()})
        for k, v in head.state_dict().items():
            out[f"cls_{tag}/w/" + k] = v.numpy()
        if tag == "b":
            head.train()
            xr = x.clone().requires_grad_(True)
            go = torch.randn(shape[0], kw["num_classes"], generator=g)
            yt = head(xr)
            (yt * go).sum().backward()
            out.update({"cls_b/go": go.numpy(), "cls_b/y_train": yt.detach().numpy(), "cls_b/dx": xr.grad.numpy()})
            for k, v in head.named_parameters():
                out["cls_b/g/" + k] = v.grad.numpy()
    np.savez(os.path.join(GOLDEN_DIR, "pointcloud.npz"), **out)
    print("wrote pointcloud", sum(v.nbytes for v in out.values() if hasattr(v, "nbytes")) // 1024, "KiB")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    if not ref_loader.reference_available():
        raise SystemExit("reference tree not available; fixtures can only be generated in the build container")
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    for name, c in ENCODER_CASES.items():
        gen_encoder_case(name, c, args.check)
    gen_tokenizers(args.check)
    gen_variants(args.check)
    gen_pointcloud(args.check)


if __name__ == "__main__":
    main()
