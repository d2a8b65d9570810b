"""CPU, world_size 2 over gloo: the data-parallel gradient path (flat buckets + one all-reduce per bucket) gives
sum-over-ranks, the buckets tile the flat buffer exactly once, and the reference-style coalesced helper averages."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from metatransformer_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(40, 96), nn.LayerNorm(96), nn.Linear(96, 33))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = _model()
        flat = parallel.FlatParams(m.parameters())
        # params / grads are views of the flat buffers
        for p, o in zip(flat.params, flat.offsets):
            assert p.data.data_ptr() == flat.flat_param.data_ptr() + 4 * o
            assert p.grad.data_ptr() == flat.flat_grad.data_ptr() + 4 * o
        x = torch.randn(16, 40, generator=torch.Generator().manual_seed(100 + rank))
        flat.zero_grad()
        m(x).square().mean().backward()
        # autograd accumulated IN PLACE into the flat views
        assert all(p.grad.data_ptr() == flat.flat_grad.data_ptr() + 4 * o for p, o in zip(flat.params, flat.offsets))
        local = flat.flat_grad.clone()
        parallel.allreduce_gradients(flat, bucket_bytes=4096)            # several small buckets
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        assert torch.allclose(flat.flat_grad, sum(gathered), atol=1e-6)
        # bucket slices cover the buffer exactly once, in reverse parameter order
        bks = flat.buckets(4096)
        assert sum(b.numel() for b in bks) == flat.numel
        assert bks[0].data_ptr() > bks[-1].data_ptr()
        # overlapped reducer: hooks fire during backward, result == plain sum, and it is reusable step after step
        red = parallel.OverlappedGradReducer(flat, bucket_bytes=4096)
        for it in range(2):
            flat.zero_grad()
            m(x).square().mean().backward()
            local2 = None
            red.finish()
            assert torch.allclose(flat.flat_grad, sum(gathered), atol=1e-6), it
        red.remove()
        # reference-style helper (dist_utils.py:14-35): average of loose tensors
        t = [torch.full((7,), float(rank + 1)), torch.full((3, 5), 10.0 * (rank + 1))]
        parallel.allreduce_coalesced(t, bucket_bytes=16)
        assert torch.allclose(t[0], torch.full((7,), 1.5)) and torch.allclose(t[1], torch.full((3, 5), 15.0))
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_flat_bucket_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _id_failure_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def boom():
            raise RuntimeError("librccl not found (simulated)")
        parallel.Comm.new_unique_id = staticmethod(boom)         # what rank 0 would see without a loadable RCCL
        try:
            parallel.Comm.from_torch_distributed(device=0)
            q.put((rank, "no error"))
        except parallel.MetaEncError as e:
            q.put((rank, "raised" if "could not draw an RCCL id" in str(e) else f"other: {e}"))
    finally:
        dist.destroy_process_group()


def test_comm_id_failure_reaches_every_rank():
    """rank 0 failing to draw the RCCL id must not leave the other ranks waiting in the broadcast: every rank raises the
    same MetaEncError (bench.py then falls back, collectively, to a torch.distributed nccl group)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_id_failure_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "raised"), (1, "raised")], res


def test_flatparams_single_process_semantics():
    m = _model()
    before = [p.detach().clone() for p in m.parameters()]
    flat = parallel.FlatParams(m.parameters())
    assert all(torch.equal(a, b) for a, b in zip(before, m.parameters())), "re-homing must not change values"
    assert flat.numel % 64 == 0
    m(torch.randn(4, 40)).sum().backward()
    assert flat.flat_grad.abs().sum() > 0
    flat.zero_grad()
    assert flat.flat_grad.abs().sum() == 0 and all(p.grad is not None for p in m.parameters())
    parallel.allreduce_gradients(flat)          # no process group: a no-op, not an error


def test_flatparams_check_repairs_or_rejects():
    """model.zero_grad() sets .grad to None, autograd then allocates fresh tensors, and a fused optimizer that
    only looks at the flat buffer would step on stale gradients.  check() (called by FusedAdamW.step) must notice."""
    from metatransformer_amd import MetaEncError
    m = _model()
    flat = parallel.FlatParams(m.parameters())
    x = torch.randn(4, 40)
    m(x).sum().backward()
    flat.check()                                            # everything aliases: fine
    want = [p.grad.clone() for p in m.parameters()]
    m.zero_grad(set_to_none=True)                           # the torch-2.x default
    with pytest.raises(MetaEncError, match="zero_grad"):
        flat.check()
    m(x).sum().backward()                                   # autograd allocates stray .grad tensors
    assert all(p.grad.data_ptr() != flat.flat_grad.data_ptr() + 4 * o for p, o in zip(flat.params, flat.offsets))
    flat.flat_grad.fill_(123.0)                             # stale content the optimizer must NOT see
    flat.check()                                            # copies the stray gradients in and re-attaches the views
    for p, o, w in zip(flat.params, flat.offsets, want):
        assert p.grad.data_ptr() == flat.flat_grad.data_ptr() + 4 * o
        assert torch.equal(p.grad, w)
    p0 = flat.params[0]
    p0.data = p0.data.clone()                               # re-homed parameter: the flat master no longer feeds the model
    with pytest.raises(MetaEncError, match="no longer lives"):
        flat.check()


def test_flatparams_no_decay_layout():
    m = _model()
    flat = parallel.FlatParams(m.named_parameters(), no_decay=parallel.no_decay_rule)
    names = {id(p): n for n, p in m.named_parameters()}
    order = [names[id(p)] for p in flat.params]
    head = [n for n in order if n.endswith(".bias") or n == "1.weight"]       # 1 = the LayerNorm
    assert order[:len(head)] == head and all(n.endswith(".weight") and n != "1.weight" for n in order[len(head):])
    assert flat.no_decay_numel == flat.offsets[len(head)] and 0 < flat.no_decay_numel < flat.numel
    m(torch.randn(4, 40)).sum().backward()                  # views still work in the permuted layout
    assert all(p.grad.data_ptr() == flat.flat_grad.data_ptr() + 4 * o for p, o in zip(flat.params, flat.offsets))


def test_reducer_accumulation_needs_no_sync():
    from metatransformer_amd import MetaEncError
    m = _model()
    flat = parallel.FlatParams(m.parameters())
    red = parallel.OverlappedGradReducer(flat, bucket_bytes=4096)
    x = torch.randn(4, 40)
    flat.zero_grad()
    with red.no_sync():
        m(x).sum().backward()                               # accumulation micro-step: nothing is counted
    m(x).sum().backward()
    red.finish()
    m(x).sum().backward()
    with pytest.raises(MetaEncError, match="no_sync"):      # a second backward before finish(): caught, not mis-reduced
        m(x).sum().backward()
    red.remove()


class _DirectLinearFn(torch.autograd.Function):
    """CPU stand-in for the fused Block backward's gradient route (encoder.py _BlockFn.backward / _backward_c): the weight
    gradient is ACCUMULATED in place into the FlatParams view (what the wgrad GEMM's beta = 1 epilogue does), announced with
    flat.grad_written, and autograd gets None for it -- while the bias gradient travels the ordinary autograd route."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.w = w
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        flat = ctx.w._me_flat
        tgt = flat.direct_grad(ctx.w)
        assert tgt is not None
        tgt.add_(dy.t() @ x)
        flat.grad_written(ctx.w)
        return dy @ w, None, dy.sum(0)


def _mixed_route_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = _model()
        flat = parallel.FlatParams(m.named_parameters(), no_decay=parallel.no_decay_rule)
        red = parallel.OverlappedGradReducer(flat, bucket_bytes=4096)
        x = torch.randn(16, 40, generator=torch.Generator().manual_seed(200 + rank))

        def fwd():
            h = _DirectLinearFn.apply(x, m[0].weight, m[0].bias)           # in-place route (weight) + hook route (bias)
            return m[2](m[1](h))                                          # ordinary autograd modules: hook route
        ref_m = _model()
        ref_m(x).square().mean().backward()
        local = torch.cat([p.grad.reshape(-1) for p in ref_m.parameters()])
        parts = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(parts, local)
        want = {n: g for (n, p), g in zip(ref_m.named_parameters(), torch.split(sum(parts), [p.numel() for p in ref_m.parameters()]))}
        for it in range(3):                                               # every parameter counted once per step, step after step
            flat.zero_grad()
            fwd().square().mean().backward()
            red.finish()
            for n, p in m.named_parameters():
                assert torch.allclose(p.grad.reshape(-1), want[n], atol=1e-6), (it, n)
        # bf16 on the wire: same sums up to the bf16 rounding of each rank's bucket
        red.remove()
        red = parallel.OverlappedGradReducer(flat, bucket_bytes=4096, wire_dtype=torch.bfloat16)
        for it in range(2):
            flat.zero_grad()
            fwd().square().mean().backward()
            red.finish()
            for n, p in m.named_parameters():
                assert p.grad.dtype == torch.float32
                assert torch.allclose(p.grad.reshape(-1), want[n], atol=2e-2 * float(want[n].abs().max()) + 1e-6), (it, n)
            assert not all(torch.equal(p.grad.reshape(-1), want[n]) for n, p in m.named_parameters())
        # a gradient that arrives in a fresh tensor next to a reducer is refused (its bucket was reduced without it)
        from metatransformer_amd import MetaEncError
        m[2].bias.grad = None
        fwd().square().mean().backward()
        red.finish()
        try:
            flat.check()
            q.put((rank, "re-homed gradient accepted"))
            return
        except MetaEncError as e:
            assert "OverlappedGradReducer" in str(e)
        red.remove()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()[-1200:] or repr(e)))
    finally:
        dist.destroy_process_group()


def test_reducer_counts_fused_inplace_and_hook_routes_once_world2():
    """world-2 semantics with the fused in-place gradient route in the loop (not only nn.Linear)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mixed_route_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` without a torchrun environment re-executes itself under torch.distributed.run."""
    import importlib.util
    import sys
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class R:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
