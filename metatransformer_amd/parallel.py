"""Batch-sharded data parallelism for the encoder: one process per GPU, flat gradient buckets, one RCCL all-reduce
per bucket over xGMI.  On GPUs the all-reduce goes through the C ABI (`Comm` = me_comm_init / me_allreduce_bucket /
me_comm_join: RCCL on a communication stream of its own, event hand-off from the backward stream); torch.distributed
is the host-side rendezvous only (it carries the 128-byte RCCL id) -- and the reduction path of the CPU tests (gloo).

Semantics follow the reference's explicit helper, Image/segmentation/mmseg_custom/core/utils/dist_utils.py:14-55
(`_allreduce_coalesced`): take tensors in buckets -> flatten -> all_reduce(sum) -> divide by world size ->
unflatten/copy back.  Here the "flatten / copy back" is free: every parameter's .grad is a VIEW into one flat fp32
buffer (SURVEY.md 8b "parameter identity": one nn.Parameter per reference tensor, grads land in each .grad), so a
bucket is just a slice of that buffer.  The 1/world scale is folded into the fused AdamW step (grad_scale).

Bucket size: xGMI is point-to-point (7 links x ~153 GB/s per GPU), ring all-reduce is per-link bound, so large buckets
amortise latency best; the default 64 MiB gives 6 buckets for Base fp32 grads (340 MB), enough to overlap the tail of
backward with communication when `overlap=True` (buckets are reduced on a side stream as soon as backward has
finished writing them, last layers first).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import ops
from ._capi import MetaEncError


class Comm:
    """RCCL communicator behind the C ABI (include/metaenc.h: me_comm_*).  One per process / GPU."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: Optional[int] = None):
        import ctypes
        from . import _capi
        self._lib = _capi.load()
        if len(unique_id) != _capi.ME_COMM_ID_BYTES:
            raise MetaEncError(f"Comm: the RCCL id must be {_capi.ME_COMM_ID_BYTES} bytes")
        dev = torch.cuda.current_device() if device is None else device
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), _capi.ME_COMM_ID_BYTES)
        _capi.check(self._lib.me_comm_init(ctypes.byref(h), buf, rank, world, dev), "me_comm_init")
        self._h, self.rank, self.world = h, rank, world
        try:        # RCCL prints a version banner through C stdio at init: push it out now, not after the caller's own output
            ctypes.CDLL(None).fflush(None)
        except Exception:       # noqa: BLE001
            pass

    @staticmethod
    def new_unique_id() -> bytes:
        import ctypes
        from . import _capi
        buf = ctypes.create_string_buffer(_capi.ME_COMM_ID_BYTES)
        _capi.check(_capi.load().me_comm_unique_id(buf), "me_comm_unique_id")
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, group=None, device: Optional[int] = None) -> "Comm":
        """Rank 0 draws the id, torch.distributed (any backend) carries it to the other ranks."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [None]
        if rank == 0:
            try:
                box[0] = cls.new_unique_id()
            except Exception as e:       # noqa: BLE001 -- the other ranks are waiting in the broadcast: tell them, then raise
                box[0] = e
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if isinstance(box[0], Exception):
            raise MetaEncError(f"Comm: rank 0 could not draw an RCCL id: {box[0]}")
        return cls(box[0], rank, world, device)

    def allreduce(self, t: torch.Tensor, producer_stream: Optional[int] = None) -> None:
        """In-place sum over ranks, enqueued on the communicator's stream behind everything already enqueued on the
        producer stream (default: torch's current stream)."""
        from . import _capi
        if not t.is_cuda or not t.is_contiguous():
            raise MetaEncError("Comm.allreduce: contiguous CUDA tensors only")
        ps = torch.cuda.current_stream().cuda_stream if producer_stream is None else producer_stream
        _capi.check(self._lib.me_allreduce_bucket(self._h, t.data_ptr(), t.numel(), _capi.dtype_code(t.dtype), ps),
                    "me_allreduce_bucket")

    def join(self, consumer_stream: Optional[int] = None) -> None:
        from . import _capi
        cs = torch.cuda.current_stream().cuda_stream if consumer_stream is None else consumer_stream
        _capi.check(self._lib.me_comm_join(self._h, cs), "me_comm_join")

    def info(self):
        import ctypes
        r, w, n = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
        from . import _capi
        _capi.check(self._lib.me_comm_info(self._h, ctypes.byref(r), ctypes.byref(w), ctypes.byref(n)), "me_comm_info")
        return {"rank": r.value, "world": w.value, "buckets_reduced": n.value}

    def destroy(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.me_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass


def no_decay_rule(name: str, p: torch.nn.Parameter) -> bool:
    """The reference fine-tuning recipes exempt 1-D parameters and biases from weight decay
    (Video/optim_factory.py:67-73: `len(param.shape) == 1 or name.endswith(".bias")`)."""
    return p.dim() == 1 or name.endswith(".bias")


def layer_id_for_vit(name: str, num_layers: int) -> int:
    """get_num_layer_for_vit (Video/optim_factory.py:28-41; the same rule keyed on `backbone.blocks.N` in
    Image/segmentation/mmcv_custom/layer_decay_optimizer_constructor.py:17-41): embeddings -> 0, block N -> N + 1, the rest
    (final norm / head) -> num_layers - 1, where num_layers = depth + 2.  Accepts the wrapped (`blocks.3.attn...`,
    `backbone.blocks.3...`) and the bare nn.Sequential (`3.attn...`) spellings of a block parameter."""
    import re
    base = name.split(".")[-1] if "." not in name else name
    if base in ("cls_token", "mask_token", "pos_embed") or name.endswith(("cls_token", "mask_token", "pos_embed")):
        return 0
    if name.startswith("patch_embed") or ".patch_embed" in name:
        return 0
    if name.startswith("rel_pos_bias"):
        return num_layers - 1
    m = re.match(r"^(?:.*\.)?blocks\.(\d+)\.", name) or re.match(r"^(\d+)\.", name)
    if m:
        return int(m.group(1)) + 1
    return num_layers - 1


def layer_decay_scales(depth: int, layer_decay: float) -> List[float]:
    """LayerDecayValueAssigner's table (Video/run_class_finetuning.py: `layer_decay ** (num_layers + 1 - i)` for
    i in range(num_layers + 2)), indexed by layer id."""
    n = depth + 2
    return [layer_decay ** (n - 1 - i) for i in range(n)]


class FlatParams:
    """Re-homes parameters (and their gradients) into flat fp32 buffers; `p.data` / `p.grad` become views.

    Use `flat.zero_grad()` (NOT model.zero_grad() / a torch optimizer's zero_grad(), which set .grad to None): the fused
    optimizer reads the flat gradient buffer.  `FusedAdamW.step()` verifies that every p.data / p.grad still aliases the
    flat buffers and repairs or rejects what does not."""

    def __init__(self, params: Iterable, fused_accumulate: bool = True, no_decay=None):
        """params: parameters, or (name, parameter) pairs (model.named_parameters()).
        fused_accumulate: let the Block backward accumulate weight gradients straight into the flat buffer (see
        direct_grad).  Turn it off if you call torch.autograd.grad() on encoder weights (they would come back None).
        no_decay: optional predicate (name, parameter) -> bool; those parameters are laid out FIRST in the flat buffers,
        [0, no_decay_numel), so that FusedAdamW applies weight decay to the rest only (two launches)."""
        self.fused_accumulate = fused_accumulate
        named = [(q if isinstance(q, tuple) else ("", q)) for q in params]
        named = [(n, p) for n, p in named if p.requires_grad]
        if no_decay is not None:
            head = [(n, p) for n, p in named if no_decay(n, p)]
            tail = [(n, p) for n, p in named if not no_decay(n, p)]
            named = head + tail
        self.params: List[torch.nn.Parameter] = [p for _, p in named]
        self.names: List[str] = [n for n, _ in named]
        if not self.params:
            raise MetaEncError("FlatParams: no trainable parameters")
        dev = self.params[0].device
        dt = self.params[0].dtype
        if any(p.dtype != dt or p.device != dev for p in self.params):
            raise MetaEncError("FlatParams: parameters must share dtype and device")
        # 64-element alignment keeps every view 256-byte aligned for the vectorised kernels
        self.offsets, off = [], 0
        self.no_decay_numel = 0
        for n, p in named:
            self.offsets.append(off)
            off += (p.numel() + 63) // 64 * 64
            if no_decay is not None and no_decay(n, p):
                self.no_decay_numel = off
        self.numel = off
        self.flat_param = torch.zeros(off, dtype=dt, device=dev)
        self.flat_grad = torch.zeros(off, dtype=dt, device=dev)
        self.flat_bf16: Optional[torch.Tensor] = None      # bf16 mirror of flat_param, written by the fused optimizer
        self._mirror_epoch = -1
        self._mirror_versions: List[int] = []
        self._index = {}
        self._listeners = []        # callables(param_index): a gradient was written straight into the flat buffer
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            self.flat_param[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat_param[o:o + p.numel()].view(p.shape)
            p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)
            self._index[id(p)] = i
            p._me_flat = self       # lets the fused Block backward accumulate weight gradients in place (direct_grad)

    def direct_grad(self, p: torch.nn.Parameter) -> Optional[torch.Tensor]:
        """The flat-buffer view that IS p.grad, or None if something replaced it.  The Block backward accumulates weight
        (and bias) gradients into it from the GEMM epilogue (beta = 1) and returns None to autograd for that parameter,
        which saves autograd's `grad += new` pass over every weight; `grad_written` then stands in for the
        post-accumulate hook."""
        i = self._index.get(id(p))
        if i is None or p.grad is None or not self.fused_accumulate:
            return None
        o = self.offsets[i]
        if p.grad.data_ptr() != self.flat_grad.data_ptr() + o * self.flat_grad.element_size() or p.grad.shape != p.shape:
            return None
        return p.grad

    def bf16_view(self, p: torch.nn.Parameter) -> Optional[torch.Tensor]:
        """bf16 copy of p out of the optimizer's mirror, or None when there is none / it is stale (the parameter was
        written by anything but the fused optimizer since: load_state_dict, manual edits -> _version moved on)."""
        i = self._index.get(id(p))
        if i is None or self.flat_bf16 is None or self._mirror_epoch != ops.WEIGHT_EPOCH or self._mirror_versions[i] != p._version:
            return None
        o = self.offsets[i]
        if p.data_ptr() != self.flat_param.data_ptr() + o * self.flat_param.element_size():
            return None
        return self.flat_bf16[o:o + p.numel()].view(p.shape)

    def grad_written(self, p: torch.nn.Parameter) -> None:
        i = self._index[id(p)]
        for cb in self._listeners:
            cb(i)

    def zero_grad(self) -> None:
        if getattr(self, "_zero_is_free", False):      # FusedAdamW(overlap=True) zeroed every slice behind its update
            self._zero_is_free = False
        else:
            self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):       # re-attach views if something replaced them
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + o * self.flat_grad.element_size():
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)

    def check(self) -> None:
        """Called by FusedAdamW.step(): every p.data / p.grad must still alias the flat buffers.
        * p.grad replaced by a fresh tensor (autograd allocates one after `model.zero_grad()` set it to None): its values
          are copied into the flat slice and the view is re-attached -- the step then sees the right gradient;
        * p.grad is None at step time, or p.data was re-homed (.to() / .cuda() / load into a new storage): MetaEncError --
          stepping would silently apply stale gradients or update a buffer the model no longer reads."""
        es = self.flat_grad.element_size()
        gbase, pbase = self.flat_grad.data_ptr(), self.flat_param.data_ptr()
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            if p.data_ptr() != pbase + o * es:
                raise MetaEncError(f"FlatParams: parameter #{i} {tuple(p.shape)} no longer lives in the flat buffer "
                                   "(moved by .to()/.cuda()/assignment after FlatParams was built); rebuild FlatParams")
            g = p.grad
            if g is None:
                raise MetaEncError(f"FlatParams: parameter #{i} {tuple(p.shape)} has .grad = None at step time -- use "
                                   "flat.zero_grad() instead of model.zero_grad() / optimizer.zero_grad(set_to_none=True)")
            if g.data_ptr() != gbase + o * es or g.shape != p.shape:
                if self._listeners:
                    # a gradient reducer watches this buffer: its bucket all-reduce was launched on the flat slice when the
                    # hook fired, i.e. WITHOUT this gradient -- re-homing it now would step every rank with its local,
                    # unreduced values and let the replicas drift apart silently
                    raise MetaEncError(f"FlatParams: parameter #{i} {tuple(p.shape)} received its gradient in a fresh tensor "
                                       "while an OverlappedGradReducer is attached (p.grad was set to None / replaced): the "
                                       "bucket all-reduce has already run without it -- use flat.zero_grad()")
                view = self.flat_grad[o:o + p.numel()].view(p.shape)
                view.copy_(g)
                p.grad = view

    def buckets(self, bucket_bytes: int) -> List[torch.Tensor]:
        """Slices of the flat gradient, split at parameter boundaries, in REVERSE parameter order (the order
        backward finishes them: last layer first)."""
        es = self.flat_grad.element_size()
        out, end = [], self.numel
        start_idx = len(self.params) - 1
        cur_end = end
        for i in range(len(self.params) - 1, -1, -1):
            if (cur_end - self.offsets[i]) * es >= bucket_bytes or i == 0:
                out.append(self.flat_grad[self.offsets[i]:cur_end])
                cur_end = self.offsets[i]
        return out


def allreduce_gradients(flat: FlatParams, group=None, bucket_bytes: int = 64 << 20, average: bool = False,
                        force: bool = False, comm: Optional[Comm] = None) -> None:
    """One all_reduce(sum) per flat bucket.  `average=True` divides by the world size afterwards (reference
    semantics, dist_utils.py:31-32); the bench leaves it False and folds 1/world into the optimizer."""
    if comm is not None:
        if comm.world == 1 and not force:
            return
        for b in flat.buckets(bucket_bytes):
            comm.allreduce(b)
        comm.join()
        if average:
            flat.flat_grad.div_(comm.world)
        return
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return
    handles = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group, async_op=True) for b in flat.buckets(bucket_bytes)]
    for h in handles:
        h.wait()
    if average:
        flat.flat_grad.div_(dist.get_world_size(group))


class OverlappedGradReducer:
    """All-reduce each flat bucket as soon as backward has finished writing it, overlapping RCCL with the rest of
    backward (what DDP does with its 25 MB buckets, Video/run_class_finetuning.py:739-742; here the buckets are slices
    of the FlatParams gradient buffer, last layers first, so nothing is copied).

    Every parameter gets a post-accumulate-grad hook; a bucket launches ``all_reduce(async_op=True)`` when its last
    parameter has fired.  ``finish()`` waits for the handles (call it before the optimizer step)."""

    def __init__(self, flat: FlatParams, group=None, bucket_bytes: int = 64 << 20, force: bool = False,
                 comm: Optional[Comm] = None, wire_dtype: Optional[torch.dtype] = None):
        """comm: a `Comm` (RCCL behind the C ABI) -- the GPU path; without one the buckets go through torch.distributed
        (`group`), which is what the CPU tests use with gloo.
        One backward per step is assumed; for gradient accumulation wrap the extra backwards in `no_sync()` (every
        reduction is then deferred to the backward that runs outside it, as DDP.no_sync does).
        wire_dtype=torch.bfloat16: a finished bucket is rounded to bf16 into a scratch buffer, THAT is all-reduced (half the
        bytes per xGMI link: 170 MB instead of 340 MB per Base step) and `finish()` converts the sums back into the fp32
        flat buffer -- the accumulation of the local gradient stays fp32 in the wgrad epilogue; only the exchange is bf16
        (what mmcv's fp16 all-reduce hook does for the reference's fp16 runs).  Default: fp32 on the wire."""
        self.flat, self.group, self.force, self.comm = flat, group, force, comm
        if wire_dtype not in (None, torch.float32, torch.bfloat16):
            raise MetaEncError("OverlappedGradReducer: wire_dtype must be None / float32 / bfloat16")
        self.wire_dtype = None if wire_dtype == torch.float32 else wire_dtype
        self._wire = None               # bf16 scratch, one slice per bucket (allocated on first use)
        self._defer = False
        es = flat.flat_grad.element_size()
        # same partition as FlatParams.buckets(): walk parameters in reverse order
        self.bucket_of, self.bucket_slices, self.bucket_left = {}, [], []
        cur_end, members = flat.numel, []
        for i in range(len(flat.params) - 1, -1, -1):
            members.append(i)
            if (cur_end - flat.offsets[i]) * es >= bucket_bytes or i == 0:
                b = len(self.bucket_slices)
                self.bucket_slices.append(flat.flat_grad[flat.offsets[i]:cur_end])
                self.bucket_left.append(len(members))
                for m in members:
                    self.bucket_of[m] = b
                cur_end, members = flat.offsets[i], []
        self._initial = list(self.bucket_left)
        self.handles = []
        # Two notification routes per parameter: autograd's post-accumulate hook, and FlatParams.grad_written for gradients
        # the fused backward accumulated in place (it hands autograd None for those).  torch still runs the hook for a
        # None gradient, so a parameter is usually announced TWICE per backward -- it is counted once, on the first
        # announcement (both come after the kernel that writes the gradient was enqueued); the same ROUTE announcing a
        # parameter twice means a second backward.
        self._seen = [0] * len(flat.params)             # bit 0: hook, bit 1: direct
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(flat.params)]
        self._direct = lambda i: self._fire(i, 2)       # gradients the fused backward wrote in place (FlatParams.direct_grad)
        flat._listeners.append(self._direct)

    def _active(self) -> bool:
        if self.comm is not None:
            return self.comm.world > 1 or self.force
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force)

    @staticmethod
    def _convert(src: torch.Tensor, dst: torch.Tensor) -> None:
        if src.is_cuda:
            ops.cast(src, dst.dtype, out=dst)
        else:
            dst.copy_(src)

    def _reduce(self, b: int, blocking: bool = False) -> None:
        buf = self.bucket_slices[b]
        if self.wire_dtype is not None:
            if self._wire is None:
                self._wire = [torch.empty(t.numel(), dtype=self.wire_dtype, device=t.device) for t in self.bucket_slices]
                self._wired = []
            self._convert(buf, self._wire[b])                   # on the producer stream, ahead of the all-reduce below
            buf = self._wire[b]
            self._wired.append(b)
        if self.comm is not None:
            self.comm.allreduce(buf)                            # asynchronous on the communicator's stream
        elif blocking:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        else:
            self.handles.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def no_sync(self):
        """Context manager for gradient accumulation: backwards inside it only accumulate locally."""
        reducer = self

        class _NoSync:
            def __enter__(self_):
                reducer._defer = True

            def __exit__(self_, *exc):
                reducer._defer = False
                reducer.bucket_left = list(reducer._initial)
                reducer._seen = [0] * len(reducer._seen)
        return _NoSync()

    def _fire(self, idx, route: int = 1) -> None:
        if self._defer:
            return
        seen = self._seen[idx]
        if seen & route:
            raise MetaEncError("OverlappedGradReducer: a parameter received a second gradient before finish() -- one "
                               "backward per step is assumed; wrap accumulation backwards in reducer.no_sync()")
        self._seen[idx] = seen | route
        if seen:
            return                      # already counted through the other route
        b = self.bucket_of[idx]
        self.bucket_left[b] -= 1
        if self.bucket_left[b] == 0 and self._active():
            self._reduce(b)

    def _make_hook(self, idx):
        def hook(param):
            self._fire(idx)
        return hook

    def finish(self) -> None:
        for h in self.handles:
            h.wait()
        self.handles = []
        if any(left != 0 for left in self.bucket_left) and self._active():
            # a parameter received no gradient this step (unused / frozen late): reduce what was not launched
            for b, left in enumerate(self.bucket_left):
                if left != 0:
                    self._reduce(b, blocking=True)
        if self.comm is not None and self._active():
            self.comm.join()            # the current (optimizer) stream waits for every bucket reduction
        if self.wire_dtype is not None and self._wire is not None:
            for b in self._wired:       # the reduced bf16 sums back into the fp32 flat buffer (behind the join)
                self._convert(self._wire[b], self.bucket_slices[b])
            self._wired = []
        self.bucket_left = list(self._initial)
        self._seen = [0] * len(self._seen)

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        if self._direct in self.flat._listeners:
            self.flat._listeners.remove(self._direct)


def allreduce_coalesced(tensors: Sequence[torch.Tensor], group=None, bucket_bytes: int = 64 << 20) -> None:
    """Generic form for gradients that do NOT live in a FlatParams (tokenizer / head parameters of the frozen-encoder
    pipelines): bucket -> flatten -> all_reduce -> /world -> copy back, exactly dist_utils.py:14-35."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for t in bucket:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        bucket, size = [], 0

    for t in tensors:
        bucket.append(t)
        size += t.numel() * t.element_size()
        if size >= bucket_bytes:
            flush()
    flush()


class FusedAdamW:
    """AdamW over a FlatParams on the device; torch.optim.AdamW semantics.

    Plain form (the default): one launch per weight-decay slice (me_adamw_step).
    Fine-tune form -- any of `lr_scale`, `max_norm`, `loss_scale` given: the reference recipes' parameter groups and gradient
    scaler in one pass, decided on the device (me_grad_stats -> me_adamw_prepare -> me_adamw_step_segments):
      * lr_scale: callable(name) -> float, e.g. ``lambda n: scales[layer_id_for_vit(n, len(scales))]`` with
        ``scales = layer_decay_scales(depth, 0.75)`` -- layer-wise lr decay (Video/optim_factory.py:28-95); weight decay per
        parameter still follows FlatParams(no_decay=...);
      * max_norm: torch.nn.utils.clip_grad_norm_ over ALL parameters of the FlatParams (Video/utils.py:391-395);
      * loss_scale: a 1-element device tensor holding the current loss scale (GradScaler semantics: gradients are divided by
        it, and a non-finite gradient SKIPS the step: parameters, moments and the step counter stay); `step()` then returns the
        device tensors (total_norm, found_inf) without synchronising -- feed found_inf to `DynamicLossScale.update`."""

    def __init__(self, flat: FlatParams, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, bf16_mirror: bool = True,
                 lr_scale=None, max_norm: float = 0.0, loss_scale: Optional[torch.Tensor] = None, overlap: bool = False,
                 grad_scale: float = 1.0, prefetch_transposes: bool = False):
        """bf16_mirror: also emit the updated parameters in bf16 (FlatParams.flat_bf16) from the same kernel; Blocks
        running in bf16 then take their forward weight copies from it instead of re-casting every weight.
        overlap (plain form, ONE backward per step, no gradient exchange between ranks): the update of a parameter UNIT (one Block's
        parameters: a contiguous slice of the decay region and one of the no-decay region) is enqueued on a side stream as soon as
        backward has written the unit's last gradient -- the HBM-bound AdamW pass (2.6 GB for Base) then runs under the rest of
        backward, which is matrix-bound -- followed by the zero-fill of that gradient slice (`flat.zero_grad()` of the next step
        becomes free) and, behind the last unit, by the refresh of the transposed weight copies the next backward needs.  `step()`
        launches whatever is left and joins.  NOT for models that run a Block more than once per backward (an encoder shared by two
        modalities, tied weights) or accumulate gradients over several backwards: the second gradient of a parameter raises
        MetaEncError (its unit has been updated behind the first).  `grad_scale` must be given HERE (the launches happen before `step()` is called);
        after `step()` the gradients read as zero.  Same arithmetic, same bits as the one-pass form (tests).
        prefetch_transposes (one-pass form): rebuild the transposed weight copies the next BACKWARD needs on a side stream right behind
        the update, i.e. under the next forward, instead of at the head of that backward (one batched launch, ~0.15 ms for Base)."""
        self.prefetch_transposes = bool(prefetch_transposes) and not overlap
        self._pf_stream = None
        if flat.flat_param.dtype != torch.float32:
            raise MetaEncError("FusedAdamW needs fp32 master parameters")
        self.flat, self.lr, self.betas, self.eps, self.wd = flat, lr, betas, eps, weight_decay
        self.bf16_mirror = bf16_mirror
        self.exp_avg = torch.zeros_like(flat.flat_param)
        self.exp_avg_sq = torch.zeros_like(flat.flat_param)
        self.t = 0
        self.max_norm, self.loss_scale = float(max_norm or 0.0), loss_scale
        self.fine_tune = lr_scale is not None or self.max_norm > 0 or loss_scale is not None
        self.groups = None
        if self.fine_tune:
            self._build_segments(lr_scale)
        self.overlap = bool(overlap)
        self.grad_scale = float(grad_scale)
        if self.overlap:
            if self.fine_tune:
                raise MetaEncError("FusedAdamW(overlap=True): the plain form only (clipping / loss scaling need the whole gradient first)")
            self._setup_overlap()

    # ---- overlap=True: per-unit updates launched from the gradient notifications ---------------------------------------------
    def _setup_overlap(self) -> None:
        f = self.flat
        dev = f.flat_param.device
        self._side = torch.cuda.Stream(device=dev)
        # unit of a parameter: its leading numeric name component ("3.attn.qkv.weight" / "blocks.3...": Block 3), else the parameter alone
        def unit_of(name: str, i: int):
            for part in name.split("."):
                if part.isdigit():
                    return ("block", int(part))
            return ("param", i)
        units = {}
        for i, name in enumerate(f.names):
            units.setdefault(unit_of(name, i), []).append(i)
        self._units = []                  # [(param indices, [(lo, hi, weight_decay), ...])]
        for key, idx in units.items():
            ranges = []
            for region_lo, region_hi, wd in ((0, f.no_decay_numel, 0.0), (f.no_decay_numel, f.numel, self.wd)):
                mine = sorted(i for i in idx if region_lo <= f.offsets[i] < region_hi)
                if not mine:
                    continue
                if mine != list(range(mine[0], mine[-1] + 1)):
                    raise MetaEncError(f"FusedAdamW(overlap=True): the parameters of unit {key} are not contiguous in the flat buffer")
                lo = f.offsets[mine[0]]
                hi = f.offsets[mine[-1] + 1] if mine[-1] + 1 < len(f.offsets) else f.numel
                hi = min(hi, region_hi)
                ranges.append((lo, hi, wd))
            self._units.append((idx, ranges))
        self._unit_of = {}
        for u, (idx, _) in enumerate(self._units):
            for i in idx:
                self._unit_of[i] = u
        self._left = [len(idx) for idx, _ in self._units]
        self._seen = [0] * len(f.params)                # bit 0: autograd's post-accumulate hook, bit 1: FlatParams.grad_written
        self._launched = [False] * len(self._units)
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(f.params)]
        f._listeners.append(lambda i: self._fire(i, 2))      # gradients the fused Block backward wrote in place
        f._zero_is_free = False

    def _make_hook(self, i):
        def hook(param):
            self._fire(i, 1)
        return hook

    def _fire(self, i: int, route: int = 1) -> None:
        seen = self._seen[i]
        if seen & route:
            # The same route twice before step(): the parameter's Block ran more than once in this backward (one encoder shared by two
            # modalities, tied weights) or a second backward was run.  The unit may already have been updated and its gradient slice
            # zeroed behind the FIRST pass: the second pass would compute dX from updated weights and leave its weight
            # gradients in the zeroed slice.  There is no way to know at the first announcement that another one follows -- refuse.
            raise MetaEncError("FusedAdamW(overlap=True): a parameter received a second gradient before step() (a Block used more than once "
                               "in one backward -- shared encoder, weight tying -- or gradient accumulation): the per-Block update has "
                               "already been launched behind the first one, parameters and gradients are now inconsistent; use "
                               "FusedAdamW(overlap=False) for such models")
        self._seen[i] = seen | route
        if seen:
            return                                      # (announced through both routes: counted once)
        u = self._unit_of[i]
        self._left[u] -= 1
        if self._left[u] == 0 and not self._launched[u]:
            self._launch_unit(u)

    def _ensure_mirror(self) -> None:
        f = self.flat
        if self.bf16_mirror and f.flat_bf16 is None:
            f.flat_bf16 = torch.empty(f.numel, dtype=torch.bfloat16, device=f.flat_param.device)
            ops.cast(f.flat_param, torch.bfloat16, out=f.flat_bf16)

    def _launch_unit(self, u: int) -> None:
        f = self.flat
        self._ensure_mirror()
        main = torch.cuda.current_stream(f.flat_param.device)
        self._side.wait_stream(main)                    # behind the kernel that wrote the unit's last gradient
        with torch.cuda.stream(self._side):
            for lo, hi, wd in self._units[u][1]:
                ops.adamw_step(f.flat_param[lo:hi], f.flat_grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], lr=self.lr,
                               betas=self.betas, eps=self.eps, weight_decay=wd, step=self.t + 1, grad_scale=self.grad_scale,
                               bf16_mirror=f.flat_bf16[lo:hi] if self.bf16_mirror else None, bump_epoch=False)
                f.flat_grad[lo:hi].zero_()              # nobody reads this slice again before the next backward writes it
        self._launched[u] = True

    def _step_overlapped(self, grad_scale: float) -> None:
        f = self.flat
        if abs(grad_scale - self.grad_scale) > 1e-12 * max(1.0, abs(grad_scale)):
            raise MetaEncError(f"FusedAdamW(overlap=True): step(grad_scale={grad_scale}) differs from the grad_scale={self.grad_scale} the "
                               "already-launched updates used; set opt.grad_scale before backward")
        for u in range(len(self._units)):
            if not self._launched[u]:                   # a unit whose gradients never arrived (frozen late / unused): its slice is zeros
                self._launch_unit(u)
        self.t += 1
        ops.weights_updated()                           # ONE epoch for the whole step: the compute copies re-derive now, not mid-backward
        if self.bf16_mirror:
            f._mirror_epoch = ops.WEIGHT_EPOCH
            f._mirror_versions = [p._version for p in f.params]
        # the forward that follows reads the new parameters (fp32 masters / bf16 mirror): it waits for the updates and zero-fills ...
        ev = torch.cuda.Event()
        ev.record(self._side)
        torch.cuda.current_stream(f.flat_param.device).wait_event(ev)
        # ... but not for the transposed copies of the next BACKWARD, rebuilt behind them on the side stream (under the next forward);
        # their first use waits for them by itself (_WeightCache.transposed)
        with torch.cuda.stream(self._side):
            from .encoder import _WeightCache
            _WeightCache.prefetch_transposed(f.flat_param.device)
        self._left = [len(idx) for idx, _ in self._units]
        self._seen = [0] * len(f.params)
        self._launched = [False] * len(self._units)
        f._zero_is_free = True                          # every gradient slice was zeroed behind its update

    def _build_segments(self, lr_scale) -> None:
        import ctypes
        from . import _capi
        f = self.flat
        segs = []                                        # (end, lr_scale, weight_decay), merged while equal
        for i, (name, p, off) in enumerate(zip(f.names, f.params, f.offsets)):
            end = f.offsets[i + 1] if i + 1 < len(f.offsets) else f.numel
            sc = float(lr_scale(name)) if lr_scale is not None else 1.0
            wd = 0.0 if off < f.no_decay_numel else float(self.wd)
            if segs and segs[-1][1] == sc and segs[-1][2] == wd:
                segs[-1] = (end, sc, wd)
            else:
                segs.append((end, sc, wd))
        arr = (_capi.AdamwSegment * len(segs))(*[_capi.AdamwSegment(e, sc, wd) for e, sc, wd in segs])
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
        self.groups = segs
        self._segments = raw.to(f.flat_param.device)
        self._ctl = torch.zeros(ctypes.sizeof(_capi.AdamwCtl) // 4, dtype=torch.float32, device=f.flat_param.device)
        self._stats = torch.zeros(2, dtype=torch.float32, device=f.flat_param.device)

    def step(self, grad_scale: float = 1.0):
        if self.overlap:
            self.flat.check()
            return self._step_overlapped(grad_scale)
        self.t += 1
        f = self.flat
        f.check()
        if self.bf16_mirror and f.flat_bf16 is None:
            # filled from the CURRENT parameters: a first step that the device skips (found_inf) writes nothing, and the
            # mirror is handed to every Block as its forward weights from here on
            f.flat_bf16 = torch.empty(f.numel, dtype=torch.bfloat16, device=f.flat_param.device)
            ops.cast(f.flat_param, torch.bfloat16, out=f.flat_bf16)
        out = None
        if self.fine_tune:
            need_stats = self.max_norm > 0 or self.loss_scale is not None
            if need_stats:
                ops.grad_stats(f.flat_grad, out=self._stats)
            ops.adamw_step_segments(f.flat_param, f.flat_grad, self.exp_avg, self.exp_avg_sq, self._segments, len(self.groups),
                                    self._ctl, lr=self.lr, betas=self.betas, eps=self.eps, grad_scale=grad_scale,
                                    stats=self._stats if need_stats else None, loss_scale=self.loss_scale, max_norm=self.max_norm,
                                    bf16_mirror=f.flat_bf16 if self.bf16_mirror else None)
            out = (self._ctl[4:5], self._ctl[5:6])       # total_norm, found_inf (device views; no synchronisation)
        else:
            # parameters exempt from weight decay (FlatParams(no_decay=...)) sit in [0, no_decay_numel): two launches
            for lo, hi, wd in ((0, f.no_decay_numel, 0.0), (f.no_decay_numel, f.numel, self.wd)):
                if hi > lo:
                    ops.adamw_step(f.flat_param[lo:hi], f.flat_grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], lr=self.lr,
                                   betas=self.betas, eps=self.eps, weight_decay=wd, step=self.t, grad_scale=grad_scale,
                                   bf16_mirror=f.flat_bf16[lo:hi] if self.bf16_mirror else None)
        if self.bf16_mirror:      # valid for this weight epoch as long as nobody else writes the parameters
            f._mirror_epoch = ops.WEIGHT_EPOCH
            f._mirror_versions = [p._version for p in f.params]
        if self.prefetch_transposes:
            dev = f.flat_param.device
            if self._pf_stream is None:
                self._pf_stream = torch.cuda.Stream(device=dev)
            self._pf_stream.wait_stream(torch.cuda.current_stream(dev))       # behind the update
            with torch.cuda.stream(self._pf_stream):
                from .encoder import _WeightCache
                _WeightCache.prefetch_transposed(dev)       # (its first user waits for the event recorded there)
        return out


class DynamicLossScale:
    """torch.cuda.amp.GradScaler's scale bookkeeping (Video/utils.py:379, 398: `self._scaler.update()`), on the device: the scale
    is a 1-element tensor handed to FusedAdamW(loss_scale=...); `update(found_inf)` grows / backs it off with torch's own
    `_amp_update_scale_` -- PyTorch plumbing, no host synchronisation.  `scale(loss)` multiplies the loss."""

    def __init__(self, device, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000):
        self.scale_t = torch.full((1,), float(init_scale), dtype=torch.float32, device=device)
        self._growth_tracker = torch.zeros(1, dtype=torch.int32, device=device)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval

    def scale(self, loss: torch.Tensor) -> torch.Tensor:
        return loss * self.scale_t.to(loss.dtype)

    def update(self, found_inf: torch.Tensor) -> None:
        torch._amp_update_scale_(self.scale_t, self._growth_tracker, found_inf.reshape(1).float(), self.growth_factor,
                                 self.backoff_factor, self.growth_interval)
