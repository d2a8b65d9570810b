"""GPU: the data-parallel exchange step through the C ABI (me_comm_* / me_allreduce_bucket = RCCL on its own stream).
One MI355X is available to the tests, so the communicator has world size 1 -- which still exercises RCCL's
ncclCommInitRank / ncclAllReduce, the stream hand-off and the reducer integration; world-2 semantics are covered on CPU
(gloo) in tests/test_parallel_cpu.py."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_comm_world1_allreduce_and_stream_order():
    from metatransformer_amd import parallel
    comm = parallel.Comm(parallel.Comm.new_unique_id(), 0, 1)
    assert comm.info() == {"rank": 0, "world": 1, "buckets_reduced": 0}
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream()
    n = 8 << 20
    for dtype in (torch.float32, torch.bfloat16):
        buf = torch.zeros(n, dtype=dtype, device=dev)
        with torch.cuda.stream(side):
            # a long producer on a side stream: the reduction must wait for it (event hand-off), and the consumer
            # stream must wait for the reduction (me_comm_join)
            for _ in range(20):
                buf.add_(1.0)
            comm.allreduce(buf)                       # producer stream = torch's current stream = side
        comm.join()                                   # consumer = the default stream
        out = buf.clone()                             # enqueued on the default stream behind the join
        torch.cuda.synchronize()
        assert torch.equal(out, torch.full_like(out, 20.0)), dtype      # world 1: sum over ranks == identity
    assert comm.info()["buckets_reduced"] == 2
    comm.destroy()


def test_reducer_uses_the_c_abi_comm():
    import metatransformer_amd as M
    from metatransformer_amd import parallel
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    enc = M.build_encoder(2, 128, 4).to(dev)
    flat = parallel.FlatParams(enc.named_parameters(), no_decay=parallel.no_decay_rule)
    comm = parallel.Comm(parallel.Comm.new_unique_id(), 0, 1)
    red = parallel.OverlappedGradReducer(flat, comm=comm, force=True, bucket_bytes=256 << 10)
    opt = parallel.FusedAdamW(flat, lr=1e-3, weight_decay=0.1)
    x = torch.randn(4, 33, 128, device=dev, requires_grad=True)
    ref = None
    for it in range(2):
        flat.zero_grad()
        enc(x).square().mean().backward()
        red.finish()
        g = flat.flat_grad.clone()
        if ref is None:
            # same step without any reducer
            enc2 = M.build_encoder(2, 128, 4).to(dev)
            enc2.load_state_dict(enc.state_dict())
            enc2(x).square().mean().backward()
            for (n1, p1), (n2, p2) in zip(enc.named_parameters(), enc2.named_parameters()):
                assert torch.allclose(p1.grad, p2.grad, atol=1e-6, rtol=1e-5), n1
            ref = g
        opt.step()
    n_buckets = len(red.bucket_slices)
    assert n_buckets >= 2 and comm.info()["buckets_reduced"] == 2 * n_buckets
    # weight decay skipped the 1-D parameters: with zero gradient a decayed weight shrinks, a bias does not
    flat.zero_grad()
    before = flat.flat_param.clone()
    opt.exp_avg.zero_(); opt.exp_avg_sq.zero_()
    opt.step()
    torch.cuda.synchronize()
    nd = flat.no_decay_numel
    assert torch.equal(flat.flat_param[:nd], before[:nd])
    assert (flat.flat_param[nd:] - before[nd:]).abs().max() > 0
    red.remove()
    comm.destroy()


def test_bench_forced_dist_on_one_gpu():
    """ME_BENCH_FORCE_DIST=1 python bench.py --gpus 1: the whole step with the RCCL-behind-the-C-ABI exchange in it."""
    env = dict(os.environ, ME_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--batch", "16", "--no-cpu-baseline", "--no-fwd-leg"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and "me_allreduce_bucket" in line["config"]["grad_allreduce"]
    assert line["roofline"]["achieved"] > 0 and line["value"] > 0


def _multirank_worker(rank, world, port, q):
    """one process per GPU: gloo rendezvous carries the RCCL id, gradients go through me_allreduce_bucket from the
    reducer's hooks (real Blocks: the fused backward accumulates weight gradients in place and announces them itself)"""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import metatransformer_amd as M
        from metatransformer_amd import parallel
        comm = parallel.Comm.from_torch_distributed()
        torch.manual_seed(0)                                   # same weights everywhere
        enc = M.build_encoder(2, 128, 4).to(dev)
        for b in enc:
            b.compute_dtype = torch.bfloat16
        flat = parallel.FlatParams(enc.named_parameters(), no_decay=parallel.no_decay_rule)
        red = parallel.OverlappedGradReducer(flat, comm=comm, bucket_bytes=256 << 10)
        x = torch.randn(4, 33, 128, generator=torch.Generator().manual_seed(100 + rank)).to(dev).bfloat16().requires_grad_(True)
        with red.no_sync():                                    # this rank's own gradient: a backward the reducer ignores
            flat.zero_grad()
            enc(x).float().square().mean().backward()
            local = flat.flat_grad.clone().cpu()
        parts = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(parts, local)
        want = sum(parts)
        for it in range(2):                                    # reusable step after step
            flat.zero_grad()
            enc(x).float().square().mean().backward()
            red.finish()
            torch.cuda.synchronize()
            err = float((flat.flat_grad.cpu() - want).abs().max() / want.abs().max())
            assert err < 1e-5, (it, err)
        assert comm.info()["world"] == world and comm.info()["buckets_reduced"] == 2 * len(red.bucket_slices)
        red.remove()
        comm.destroy()
        q.put((rank, "ok"))
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()[-1500:] or repr(e)))
    finally:
        dist.destroy_process_group()


def test_multirank_reducer_sums_over_every_visible_gpu():
    """world = torch.cuda.device_count() ranks through the C-ABI RCCL path (skipped on the 1-GPU test box; on a multi-GPU
    node it is the first thing that exercises me_comm_init(world > 1) outside bench.py)"""
    import socket
    import torch.multiprocessing as mp
    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip("needs >= 2 visible GPUs (one rank per GPU)")
    world = min(world, 8)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_multirank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def test_bf16_wire_reducer_round_trip_on_the_gpu():
    """OverlappedGradReducer(wire_dtype=bfloat16) through me_allreduce_bucket(ME_BF16) with a world of one: fp32 bucket -> me_cast
    -> bf16 all-reduce (identity) -> me_cast back behind the join.  The result is the fp32 gradient rounded to bf16 once (relative
    error <= 2^-9 per element, exactly representable values untouched), and the optimizer stream sees it only after the join."""
    import metatransformer_amd as M
    from metatransformer_amd import parallel
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    enc = M.build_encoder(2, 128, 4).to(dev)
    flat = parallel.FlatParams(enc.named_parameters(), no_decay=parallel.no_decay_rule)
    comm = parallel.Comm(parallel.Comm.new_unique_id(), 0, 1)
    red = parallel.OverlappedGradReducer(flat, comm=comm, force=True, bucket_bytes=256 << 10, wire_dtype=torch.bfloat16)
    x = torch.randn(4, 33, 128, device=dev, requires_grad=True)
    with red.no_sync():
        flat.zero_grad()
        enc(x).square().mean().backward()
        want32 = flat.flat_grad.clone()
    for it in range(2):
        flat.zero_grad()
        enc(x).square().mean().backward()
        red.finish()
        got = flat.flat_grad.clone()
        torch.cuda.synchronize()
        assert torch.equal(got, want32.bfloat16().float()), it          # one bf16 rounding of every element, nothing else
        assert float((got - want32).abs().max()) <= 2.0 ** -8 * float(want32.abs().max())
    assert comm.info()["buckets_reduced"] == 2 * len(red.bucket_slices)
    red.remove()
    comm.destroy()


def _ddp_worker(port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1)
        import metatransformer_amd as M
        torch.manual_seed(0)
        enc = M.build_encoder(2, 128, 4).to(dev)
        ref = M.build_encoder(2, 128, 4).to(dev)
        ref.load_state_dict(enc.state_dict())
        # the wrapper the reference's fine-tune scripts put around the model (Video/run_class_finetuning.py:739-742,
        # PointCloud/examples/classification/train.py:83-87): its reducer hooks the .grad of ordinary nn.Parameters that a custom
        # autograd.Function fills
        ddp = torch.nn.parallel.DistributedDataParallel(enc, device_ids=[0])
        x = torch.randn(4, 33, 128, generator=torch.Generator().manual_seed(5)).to(dev)
        res = {}
        for name, model in (("ddp", ddp), ("plain", ref)):
            for amp in (False, True):
                model.zero_grad(set_to_none=True)
                xr = x.clone().requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                    y = model(xr)
                y.float().square().mean().backward()
                mod = model.module if name == "ddp" else model
                res[(name, amp)] = (y.detach().clone(), xr.grad.clone(), {k: p.grad.clone() for k, p in mod.named_parameters()})
        for amp in (False, True):
            (ya, dxa, ga), (yb, dxb, gb) = res[("ddp", amp)], res[("plain", amp)]
            assert torch.equal(ya, yb) and torch.equal(dxa, dxb), amp
            for k in ga:
                assert torch.equal(ga[k], gb[k]), (k, amp)       # world 1: averaged over one rank == the local gradient, bit for bit
        # a frozen encoder under DDP with a trainable tokenizer in front (the common reference set-up): only dL/dx flows
        for p in enc.parameters():
            p.requires_grad_(False)
            p.grad = None
        tok = torch.nn.Linear(16, 128).to(dev)
        stack = torch.nn.parallel.DistributedDataParallel(torch.nn.Sequential(tok, enc), device_ids=[0])
        stack(torch.randn(4, 33, 16, device=dev)).square().mean().backward()
        assert tok.weight.grad is not None and all(p.grad is None for p in enc.parameters())
        q.put("ok")
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put(traceback.format_exc()[-2500:] or repr(e))
    finally:
        try:
            dist.destroy_process_group()
        except Exception:       # noqa: BLE001
            pass


def test_blocks_inside_distributed_data_parallel_world1():
    """SURVEY 8b lists DistributedDataParallel among the wrappers the drop-in must survive: a 2-block encoder wrapped in DDP on a
    world-1 nccl (= RCCL) group gives the same outputs and gradients as the unwrapped encoder, fp32 and under bf16 autocast."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_ddp_worker, args=(29000 + os.getpid() % 2000, q))
    p.start()
    try:
        msg = q.get(timeout=300)
    finally:
        p.join(60)
        if p.is_alive():
            p.kill()
    assert msg == "ok", msg


def test_gradients_bit_identical_while_a_communication_kernel_holds_cus():
    """Rehearsal of multi-GPU CU contention on one GPU (tools/cu_hog.hip, tools/contention.py): while backward runs, a kernel on
    another stream holds 16 / 40 CUs the way an RCCL all-reduce would (launched from the reducer's bucket hooks, optimizer joins).
    The resident NT GEMMs and the persistent attention backward claim their work from counters, the weight-gradient GEMMs run on
    me_block_bwd's side stream: whatever the schedule, every gradient must equal the undisturbed run bit for bit (no kernel's
    arithmetic depends on which CUs it got), and the step must terminate (claimed work, no workgroup waiting for a CU it cannot get)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import contention
    if not os.path.isfile(contention.HOG):
        pytest.skip("tools/_build/libcuhog.so not built (python __graft_entry__.py builds it)")
    from metatransformer_amd import parallel, ops
    dev = torch.device("cuda:0")
    lib = contention.load_hog()
    results = {}
    for R in (0, 16, 40):
        comm = contention.HogComm(lib, R, 20.0, dev)          # 20 GB/s: every bucket's hold outlasts the rest of backward
        step, flat = contention.build_step(dev, comm, B=128, N=197, L=3)
        step()                                                # (first step: weight copies, counters, lazy attributes)
        flat.zero_grad()
        # one more backward WITHOUT the optimizer so that the gradient buffer is what is compared
        import metatransformer_amd as M
        torch.manual_seed(0)
        enc = M.build_encoder(3, 768, 12).to(dev)
        for p in enc.parameters():
            if p.dim() == 2:
                torch.nn.init.normal_(p, std=0.02)
        for blk in enc:
            blk.compute_dtype = torch.bfloat16
        enc.train()
        fl = parallel.FlatParams(enc.named_parameters())
        red = parallel.OverlappedGradReducer(fl, comm=comm, force=True, bucket_bytes=8 << 20)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(128, 197, 768, generator=g).to(dev).bfloat16().requires_grad_(True)
        gy = (torch.randn(128, 197, 768, generator=g) / 1000).to(dev).bfloat16()
        fl.zero_grad()
        n0 = comm.launched
        enc(x).backward(gy)
        red.finish()
        torch.cuda.synchronize()
        assert comm.launched - n0 >= 2                        # the hog really ran beside backward (several buckets)
        results[R] = (fl.flat_grad.clone(), x.grad.clone())
        red.remove()
    for R in (16, 40):
        assert torch.equal(results[R][0], results[0][0]) and torch.equal(results[R][1], results[0][1]), R


def test_gradients_reproducible_under_a_cu_reservation_and_a_hog():
    """Round 6: with me_gemm_reserve_cus(16) (what me_comm_init sets for world > 1) the weight gradients are planned as 240 balanced static
    parts (gemm_g3tn_sk_kernel).  Parts are a fixed K range and a fixed slab each, so the gradients are bit-identical run to run and with or
    without a kernel holding CUs beside backward -- and equal the one-item-per-CU plan's up to fp32 summation order."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import contention
    if not os.path.isfile(contention.HOG):
        pytest.skip("tools/_build/libcuhog.so not built (python __graft_entry__.py builds it)")
    import metatransformer_amd as M
    from metatransformer_amd import _capi, parallel
    dev = torch.device("cuda:0")
    lib, capi = contention.load_hog(), _capi.load()

    def grads(reserve, R):
        prev = capi.me_gemm_reserve_cus(reserve)
        try:
            comm = contention.HogComm(lib, R, 20.0, dev)
            torch.manual_seed(0)
            enc = M.build_encoder(2, 768, 12).to(dev)
            for p in enc.parameters():
                if p.dim() == 2:
                    torch.nn.init.normal_(p, std=0.02)
            for blk in enc:
                blk.compute_dtype = torch.bfloat16
            enc.train()
            fl = parallel.FlatParams(enc.named_parameters())
            red = parallel.OverlappedGradReducer(fl, comm=comm, force=True, bucket_bytes=8 << 20)
            g = torch.Generator().manual_seed(5)
            x = torch.randn(256, 197, 768, generator=g).to(dev).bfloat16().requires_grad_(True)
            gy = (torch.randn(256, 197, 768, generator=g) / 1000).to(dev).bfloat16()
            out = []
            for _ in range(2):
                fl.zero_grad()
                x.grad = None
                enc(x).backward(gy)
                red.finish()
                torch.cuda.synchronize()
                out.append((fl.flat_grad.clone(), x.grad.clone()))
            red.remove()
            return out
        finally:
            capi.me_gemm_reserve_cus(prev)

    base = grads(0, 0)
    quiet = grads(16, 0)
    hogged = grads(16, 16)
    assert torch.equal(quiet[0][0], quiet[1][0])                                  # run to run
    for a, b in zip(quiet, hogged):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])                # with a kernel holding 16 CUs beside backward
    assert torch.equal(base[0][1], quiet[0][1])                                   # dL/dx does not depend on the weight-gradient plan
    err = float((quiet[0][0] - base[0][0]).abs().max() / base[0][0].abs().max())
    assert 0.0 < err < 1e-5, err                                                  # same sums, another (fixed) order
