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
    "small_hd64":  dict(depth=2, dim=128, heads=2, B=3, N=37, eps=1e-5, seed=12, tok_stride=1, backward=True),
    "base_1blk":   dict(depth=1, dim=768, heads=12, B=2, N=197, eps=1e-5, seed=13, tok_stride=8, backward=True),
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


if __name__ == "__main__":
    main()
