"""GPU: the fine-tune recipe of the fused optimizer against torch's own pieces on the same tensors:
parameter groups with layer-wise lr decay and no-decay groups as Video/optim_factory.py:28-95 builds them ->
torch.optim.AdamW(param_groups); torch.nn.utils.clip_grad_norm_; GradScaler's unscale / skipped step / scale update
(Video/utils.py:376-404)."""
import pytest
import torch

from conftest import rel_err
from metatransformer_amd import ops, parallel
from metatransformer_amd._capi import MetaEncError as _MetaEncError

pytestmark = pytest.mark.gpu


def _named_params(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = {"pos_embed": (1, 10, 64), "patch_embed.proj.weight": (64, 48), "patch_embed.proj.bias": (64,)}
    for i in range(3):
        shapes.update({f"blocks.{i}.norm1.weight": (64,), f"blocks.{i}.norm1.bias": (64,), f"blocks.{i}.attn.qkv.weight": (192, 64),
                       f"blocks.{i}.attn.qkv.bias": (192,), f"blocks.{i}.mlp.fc1.weight": (256, 64), f"blocks.{i}.mlp.fc1.bias": (256,)})
    shapes.update({"fc_norm.weight": (64,), "fc_norm.bias": (64,), "head.weight": (40, 64), "head.bias": (40,)})
    return [(n, torch.nn.Parameter(torch.randn(*s, generator=g).to(dev))) for n, s in shapes.items()]


def _torch_groups(named, lr, wd, scales):
    """get_parameter_groups (Video/optim_factory.py:56-95): group = (layer id, decay / no_decay), lr_scale per layer"""
    groups = {}
    for n, p in named:
        nd = p.dim() == 1 or n.endswith(".bias")
        lid = parallel.layer_id_for_vit(n, len(scales))
        g = groups.setdefault((lid, nd), {"params": [], "weight_decay": 0.0 if nd else wd, "lr": lr * scales[lid]})
        g["params"].append(p)
    return list(groups.values())


def test_grad_stats_norm_and_nonfinite_count(dev):
    g = torch.randn(1_000_003, generator=torch.Generator().manual_seed(1)).to(dev)
    buf = torch.zeros(1_000_064, device=dev)[:1_000_003]
    buf.copy_(g)
    st = ops.grad_stats(buf)
    want = float(g.double().pow(2).sum().sqrt())
    assert abs(float(st[0]) - want) < 1e-6 * want and float(st[1]) == 0.0
    big = ops.grad_stats(buf * 1e25)                                # squares far outside fp32: still a finite norm, nothing "bad"
    assert abs(float(big[0]) / 1e25 - want) < 1e-5 * want and float(big[1]) == 0.0
    buf[12345] = float("inf"); buf[999_999] = float("nan"); buf[1_000_002] = float("-inf")
    assert float(ops.grad_stats(buf)[1]) == 3.0


@pytest.mark.parametrize("max_norm", [0.0, 0.7])
def test_fused_finetune_step_matches_torch_param_groups(dev, max_norm):
    lr, wd, betas, eps = 2e-3, 0.05, (0.9, 0.95), 1e-8
    scales = parallel.layer_decay_scales(3, 0.75)
    named = _named_params(dev)
    ref = [(n, torch.nn.Parameter(p.detach().clone())) for n, p in named]
    topt = torch.optim.AdamW(_torch_groups(ref, lr, wd, scales), betas=betas, eps=eps)
    flat = parallel.FlatParams(named, no_decay=parallel.no_decay_rule)
    opt = parallel.FusedAdamW(flat, lr=lr, betas=betas, eps=eps, weight_decay=wd, bf16_mirror=False, max_norm=max_norm,
                              lr_scale=lambda n: scales[parallel.layer_id_for_vit(n, len(scales))])
    assert len(opt.groups) >= 6                      # several (lr scale, weight decay) runs in the flat order
    g = torch.Generator().manual_seed(5)
    for step in range(3):
        flat.zero_grad()
        for (n, p), (_, q) in zip(named, ref):
            gr = torch.randn(p.shape, generator=g).to(dev) * (3.0 if step == 1 else 0.05)      # step 1 is clipped hard
            p.grad.copy_(gr)
            q.grad = gr.clone()
        if max_norm > 0:
            want_norm = torch.nn.utils.clip_grad_norm_([q for _, q in ref], max_norm)
        topt.step()
        out = opt.step()
        if max_norm > 0:
            assert abs(float(out[0]) - float(want_norm)) < 1e-4 * float(want_norm)
            assert float(out[1]) == 0.0
        for (n, p), (_, q) in zip(named, ref):
            assert rel_err(p, q) < 2e-6, (step, n)


def test_found_inf_skips_the_step_and_the_scale_backs_off(dev):
    lr, wd = 1e-2, 0.1
    named = _named_params(dev, seed=3)
    ref = [(n, torch.nn.Parameter(p.detach().clone())) for n, p in named]
    topt = torch.optim.AdamW([{"params": [q for n, q in ref if q.dim() == 1 or n.endswith(".bias")], "weight_decay": 0.0},
                              {"params": [q for n, q in ref if not (q.dim() == 1 or n.endswith(".bias"))], "weight_decay": wd}], lr=lr)
    flat = parallel.FlatParams(named, no_decay=parallel.no_decay_rule)
    scaler = parallel.DynamicLossScale(dev, init_scale=1024.0, growth_interval=1000)
    opt = parallel.FusedAdamW(flat, lr=lr, weight_decay=wd, bf16_mirror=False, loss_scale=scaler.scale_t)
    g = torch.Generator().manual_seed(9)

    def one_step(poison):
        flat.zero_grad()
        for (n, p), (_, q) in zip(named, ref):
            gr = torch.randn(p.shape, generator=g).to(dev)
            p.grad.copy_(gr * scaler.scale_t)                       # what backward of (loss * scale) leaves behind
            q.grad = gr.clone()
        if poison:
            named[4][1].grad.view(-1)[7] = float("inf")
        norm, found = opt.step(grad_scale=1.0)
        scaler.update(found)
        return float(found)

    assert one_step(False) == 0.0
    topt.step()
    before = [p.detach().clone() for _, p in named]
    m_before = opt.exp_avg.clone()
    assert one_step(True) == 1.0                                     # skipped: nothing moves, the scale halves
    assert all(torch.equal(a, p.detach()) for a, (_, p) in zip(before, named)) and torch.equal(m_before, opt.exp_avg)
    assert float(scaler.scale_t) == 512.0
    assert one_step(False) == 0.0                                    # the step counter did not advance on the skipped step
    topt.step()
    for (n, p), (_, q) in zip(named, ref):
        assert rel_err(p, q) < 2e-6, n


def test_first_step_skipped_leaves_a_valid_bf16_mirror(dev):
    """A found-inf skip on the very FIRST step writes nothing -- the bf16 mirror the Blocks then read as their forward weights
    must already hold the current parameters (it is filled when it is allocated), not uninitialised memory."""
    named = _named_params(dev, seed=11)
    flat = parallel.FlatParams(named, no_decay=parallel.no_decay_rule)
    scaler = parallel.DynamicLossScale(dev, init_scale=256.0, growth_interval=1000)
    opt = parallel.FusedAdamW(flat, lr=1e-2, weight_decay=0.1, bf16_mirror=True, loss_scale=scaler.scale_t)
    flat.zero_grad()
    for _, p in named:
        p.grad.fill_(1.0)
    named[2][1].grad.view(-1)[0] = float("nan")
    before = flat.flat_param.detach().clone()
    _, found = opt.step()
    assert float(found) == 1.0 and torch.equal(before, flat.flat_param.detach())
    assert flat.flat_bf16 is not None and torch.equal(flat.flat_bf16, before.to(torch.bfloat16))
    # a step that IS taken then rewrites it from the updated parameters
    flat.zero_grad()
    for _, p in named:
        p.grad.fill_(256.0)
    _, found = opt.step()
    assert float(found) == 0.0 and not torch.equal(before, flat.flat_param.detach())
    assert torch.equal(flat.flat_bf16, flat.flat_param.detach().to(torch.bfloat16))


def test_large_loss_scale_does_not_fake_a_nonfinite_gradient(dev):
    """GradScaler.unscale_ divides first and only then looks for inf: finite gradients whose SCALED sum of squares leaves the
    fp32 range (|g * scale| ~ 1e19 and up) must still take the step, with the unscaled norm reported."""
    named = _named_params(dev, seed=13)
    ref = [(n, torch.nn.Parameter(p.detach().clone())) for n, p in named]
    topt = torch.optim.AdamW([{"params": [q for n, q in ref if q.dim() == 1 or n.endswith(".bias")], "weight_decay": 0.0},
                              {"params": [q for n, q in ref if not (q.dim() == 1 or n.endswith(".bias"))], "weight_decay": 0.1}], lr=1e-2)
    flat = parallel.FlatParams(named, no_decay=parallel.no_decay_rule)
    scale = torch.full((1,), 2.0 ** 70, device=dev)
    opt = parallel.FusedAdamW(flat, lr=1e-2, weight_decay=0.1, bf16_mirror=False, loss_scale=scale, max_norm=1e9)
    g = torch.Generator().manual_seed(17)
    flat.zero_grad()
    for (n, p), (_, q) in zip(named, ref):
        gr = torch.randn(p.shape, generator=g).to(dev)
        p.grad.copy_(gr * scale)                                    # ~1e21 per element: squares overflow fp32, the values do not
        q.grad = gr.clone()
    want = torch.nn.utils.clip_grad_norm_([q for _, q in ref], 1e9)
    topt.step()
    norm, found = opt.step()
    assert float(found) == 0.0
    assert abs(float(norm) - float(want)) < 1e-4 * float(want)
    for (n, p), (_, q) in zip(named, ref):
        assert rel_err(p, q) < 2e-6, n


def test_overlapped_optimizer_equals_the_one_pass_form(dev):
    """FusedAdamW(overlap=True): per-Block updates launched from the gradient notifications on a side stream (under the rest of
    backward), gradient slices zeroed behind them, transposed weight copies rebuilt under the next forward.  Over several training
    steps of a real Block stack (fused in-place weight gradients, bf16 compute): parameters, both moments, the bf16 mirror and the
    loss trajectory are BIT-identical to the one-pass optimizer; the gradients read as zero after step() and zero_grad() is free."""
    import metatransformer_amd as M
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(8, 50, 256, generator=g)
    gy = torch.randn(8, 50, 256, generator=g) / 400

    def run(overlap):
        torch.manual_seed(0)
        enc = M.build_encoder(3, 256, 4).to(dev)
        for p in enc.parameters():
            if p.dim() == 2:
                torch.nn.init.normal_(p, std=0.02)
        for b in enc:
            b.compute_dtype = torch.bfloat16
        enc.train()
        flat = parallel.FlatParams(enc.named_parameters(), no_decay=parallel.no_decay_rule)
        opt = parallel.FusedAdamW(flat, lr=3e-3, weight_decay=0.05, overlap=overlap, grad_scale=0.5)
        x = x0.to(dev).bfloat16().requires_grad_(True)
        losses = []
        for _ in range(4):
            flat.zero_grad()
            x.grad = None
            y = enc(x)
            losses.append(float((y.detach().float() * gy.to(dev)).sum()))
            y.backward(gy.to(dev).bfloat16())
            opt.step(grad_scale=0.5)
            if overlap:
                torch.cuda.synchronize()
                assert float(flat.flat_grad.abs().max()) == 0.0 and flat._zero_is_free
        torch.cuda.synchronize()
        return flat.flat_param.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), flat.flat_bf16.clone(), losses, x.grad.clone()

    a, b = run(False), run(True)
    for i, name in enumerate(("parameters", "exp_avg", "exp_avg_sq", "bf16 mirror")):
        assert torch.equal(a[i], b[i]), name
    assert a[4] == b[4] and torch.equal(a[5], b[5])
    assert a[4][0] != a[4][-1]                              # (the weights really moved)
    with pytest.raises(_MetaEncError):
        parallel.FusedAdamW(parallel.FlatParams([torch.nn.Parameter(torch.zeros(8, device=dev))]), overlap=True, max_norm=1.0)
