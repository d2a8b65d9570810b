"""ctypes binding of libmetaenc.so (the C ABI declared in include/metaenc.h).

The north star asks for a "thin C-ABI cffi layer"; ``cffi`` is not installed in this image, so the same
C ABI is bound with the standard library's ``ctypes`` (no compile-time dependency, identical symbols).

The shared library is built in-tree (``python -m metatransformer_amd.build`` or ``__graft_entry__.build()``)
and MUST be present: there is no CPU or PyTorch fallback for any op -- loading fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import torch  # noqa: F401  (imported first so libamdhip64.so.7 resolves to the copy torch already loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmetaenc.so")

ME_F32, ME_BF16, ME_F16 = 0, 1, 2      # ME_F16: storage dtype of me_cast / me_transpose_cast only
ME_BF16X3 = 3                          # an fp32 matrix as three bf16 planes [hi | lo | hi] / [hi | hi | lo] (include/metaenc.h)
ME_BF16X2 = 4                          # ... as two planes [hi | lo] (me_gemm output; as an A operand with GemmDesc.a_wrap_k)
ME_GG8 = 5                             # gelu'(h) in eight bits (preact of a SAVE_GELU_GRAD GEMM / aux of an AUX_IS_FACTOR GEMM)
ME_GEMM_NT, ME_GEMM_TN = 0, 1
ME_ACT_NONE, ME_ACT_GELU = 0, 1
ME_GEMM_SAVE_GELU_GRAD, ME_GEMM_AUX_IS_FACTOR = 1, 2
ME_PROF_LN_FWD, ME_PROF_LN_BWD, ME_PROF_ATTN_FWD, ME_PROF_ATTN_BWD, ME_PROF_ROW_STATS = 16, 17, 18, 19, 20      # me_gemm_profile_rec.op codes
ME_COMM_ID_BYTES = 128
ME_RESIZE_BILINEAR, ME_RESIZE_BICUBIC = 0, 1
ME_POOL_MEAN, ME_POOL_MAX, ME_POOL_FIRST = 0, 1, 2


class MetaEncError(RuntimeError):
    pass


class GemmDesc(ctypes.Structure):
    """Mirror of ``struct me_gemm_desc`` (include/metaenc.h)."""
    _fields_ = [
        ("op", c_int32), ("ab_dtype", c_int32),
        ("M", c_int64), ("N", c_int64), ("K", c_int64),
        ("A", c_void_p), ("lda", c_int64),
        ("B", c_void_p), ("ldb", c_int64),
        ("C", c_void_p), ("ldc", c_int64), ("c_dtype", c_int32),
        ("act", c_int32),
        ("alpha", c_float), ("beta", c_float),
        ("bias", c_void_p),
        ("colscale", c_void_p),
        ("preact", c_void_p), ("ldpre", c_int64), ("preact_dtype", c_int32),
        ("aux_dtype", c_int32),
        ("aux", c_void_p), ("ldaux", c_int64),
        ("residual", c_void_p), ("ldres", c_int64), ("res_dtype", c_int32),
        ("flags", c_int32),
        ("res_row_mod", c_int64),
        ("out_group_rows", c_int64), ("out_group_stride", c_int64), ("out_row_offset", c_int64),
        ("workspace", c_void_p), ("workspace_bytes", c_int64),
        ("colsum_a", c_void_p),
        ("row_affine", c_void_p), ("col_shift", c_void_p),
        ("row_stats", c_void_p),
        ("row_parts", c_void_p), ("row_nparts", c_int32), ("row_eps", c_float),
        ("a_wrap_k", c_int64),
    ]


class PatchEmbedDesc(ctypes.Structure):
    """Mirror of ``struct me_patch_embed_desc`` (include/metaenc.h)."""
    _fields_ = ([("x", c_void_p), ("x_dtype", c_int32)]
                + [(n, c_int32) for n in ("B", "Cin", "T", "H", "W", "kt", "kh", "kw", "st", "sh", "sw")]
                + [("weight", c_void_p), ("w_dtype", c_int32), ("Cout", c_int32), ("bias", c_void_p),
                   ("pos", c_void_p), ("pos_dtype", c_int32), ("ld_pos", c_int64), ("prefix_rows", c_int32),
                   ("out", c_void_p), ("out_dtype", c_int32), ("ld_out", c_int64),
                   ("workspace", c_void_p), ("workspace_bytes", c_int64)])


class GemmProfileRec(ctypes.Structure):
    """Mirror of ``struct me_gemm_profile_rec``."""
    _fields_ = [("op", c_int32), ("ab_dtype", c_int32), ("M", c_int64), ("N", c_int64), ("K", c_int64), ("ms", c_float),
                ("plan", c_int32)]


ME_TC_BATCH = 48


class _TcItem(ctypes.Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("rows", c_int64), ("cols", c_int64)]


class TcBatch(ctypes.Structure):
    """Mirror of ``struct me_tc_batch``."""
    _fields_ = [("n", c_int32), ("src_dtype", c_int32), ("dst_dtype", c_int32), ("reserved", c_int32),
                ("item", _TcItem * ME_TC_BATCH)]


class BlockDesc(ctypes.Structure):
    """Mirror of ``struct me_block_desc`` (include/metaenc.h)."""
    _fields_ = ([("dtype", c_int32), ("res_dtype", c_int32), ("B", c_int32), ("N", c_int32), ("C", c_int32),
                 ("heads", c_int32), ("hidden", c_int32), ("eps", c_float), ("scale", c_float)]
                + [(n, c_void_p) for n in ("qkv_w", "proj_w", "fc1_w", "fc2_w", "qkv_wt", "proj_wt", "fc1_wt", "fc2_wt",
                                           "ln1_g", "ln1_b", "ln2_g", "ln2_b", "qkv_b", "proj_b", "fc1_b", "fc2_b",
                                           "gamma1", "gamma2", "qkv_wf", "fc1_wf", "qkv_s", "qkv_c", "fc1_s", "fc1_c",
                                           "x_stats", "y_stats", "x_parts", "y_parts")])


class BlockGrads(ctypes.Structure):
    """Mirror of ``struct me_block_grads``."""
    _fields_ = ([(n, c_void_p) for n in ("qkv_w", "proj_w", "fc1_w", "fc2_w", "qkv_b", "proj_b", "fc1_b", "fc2_b",
                                         "ln1_g", "ln1_b", "ln2_g", "ln2_b")]
                + [("w_dtype", c_int32), ("accumulate", c_int32)])


class AdamwSegment(ctypes.Structure):
    """Mirror of ``struct me_adamw_segment``: the flat bucket's elements [previous end, end) share one lr scale / weight decay."""
    _fields_ = [("end", c_int64), ("lr_scale", c_float), ("weight_decay", c_float)]


class AdamwCtl(ctypes.Structure):
    """Mirror of ``struct me_adamw_ctl`` (device-resident control block of the fused fine-tune step)."""
    _fields_ = [("grad_mul", c_float), ("skip", c_float), ("bc1", c_float), ("bc2_sqrt", c_float), ("total_norm", c_float),
                ("found_inf", c_float), ("step", c_int32), ("reserved", c_int32)]


# name -> (restype, argtypes).  Every symbol include/metaenc.h declares must be listed here
# (tests/test_boundary.py cross-checks the header against this table and against the built .so).
SIGNATURES = {
    "me_abi_version": (c_int, []),
    "me_last_error": (c_char_p, []),
    "me_build_arch": (c_char_p, []),
    "me_device_info": (c_int, [c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), c_char_p, c_int]),
    "me_row_stats": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "me_row_stats_partial_bytes": (c_size_t, [c_int64, c_int]),
    "me_row_stats_combine": (c_int, [c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p]),
    "me_layernorm_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                 c_int64, c_int, c_float, c_void_p]),
    "me_layernorm_bwd_workspace": (c_size_t, [c_int]),
    "me_layernorm_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                 c_int64, c_int, c_void_p, c_void_p]),
    "me_gemm_workspace_bytes": (c_size_t, [POINTER(GemmDesc)]),
    "me_gemm_fuses_colsum": (c_int, [POINTER(GemmDesc)]),
    "me_gemm_emits_row_stats": (c_int, [POINTER(GemmDesc)]),
    "me_gemm_takes_row_parts": (c_int, [POINTER(GemmDesc)]),
    "me_gemm_reserve_cus": (c_int, [c_int]),
    "me_gemm_takes_a_wrap": (c_int, [POINTER(GemmDesc)]),
    "me_gemm_takes_gg8": (c_int, [POINTER(GemmDesc)]),
    "me_gemm": (c_int, [POINTER(GemmDesc), c_void_p]),
    "me_gemm_profile_enable": (c_int, [c_int]),
    "me_gemm_profile_read": (c_int, [POINTER(GemmProfileRec), c_int]),
    "me_colsum_workspace": (c_size_t, [c_int64]),
    "me_colsum": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    "me_colsum_mul": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_int,
                              c_void_p, c_void_p]),
    "me_attention_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int,
                                 c_float, c_int, c_float, ctypes.c_uint64, c_void_p]),
    "me_attention_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                 c_void_p, c_int64, c_int, c_int, c_int, c_int, c_float, c_int, c_float, ctypes.c_uint64,
                                 c_void_p]),
    "me_attention_fp8_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "me_attention_fwd_fp8": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                     c_void_p, c_size_t, c_void_p]),
    "me_block_saved_bytes": (c_size_t, [POINTER(BlockDesc)]),
    "me_block_emits_stats": (c_int, [POINTER(BlockDesc)]),
    "me_block_workspace_bytes": (c_size_t, [POINTER(BlockDesc), c_int]),
    "me_block_fwd": (c_int, [POINTER(BlockDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "me_encoder_fwd": (c_int, [POINTER(BlockDesc), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "me_block_bwd": (c_int, [POINTER(BlockDesc), c_void_p, c_void_p, c_void_p, c_void_p, POINTER(BlockGrads), c_void_p,
                             c_size_t, c_void_p]),
    "me_block_bwd_overlap": (c_int, [c_int]),
    "me_cast": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "me_attention_fwd_x3": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "me_attention_bwd_x3": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                    c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "me_split3": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "me_transpose_cast": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int64, c_void_p]),
    "me_transpose_cast_batched": (c_int, [POINTER(TcBatch), c_void_p]),
    "me_split3_batched": (c_int, [POINTER(TcBatch), c_int, c_int, c_void_p]),
    "me_add_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int64, c_int64, c_int, c_void_p]),
    "me_window_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "me_dropout_add": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int64, c_float, c_float,
                               ctypes.c_uint64, c_void_p, c_void_p]),
    "me_patchify": (c_int, [c_void_p, c_int, c_void_p, c_int] + [c_int] * 11 + [c_void_p]),
    "me_unpatchify_add": (c_int, [c_void_p, c_int, c_void_p] + [c_int] * 11 + [c_void_p]),
    "me_patch_embed_fused": (c_int, [POINTER(PatchEmbedDesc)]),
    "me_patch_embed_workspace_bytes": (c_size_t, [POINTER(PatchEmbedDesc)]),
    "me_patch_embed": (c_int, [POINTER(PatchEmbedDesc), c_void_p]),
    "me_patch_embed_wgrad_fused": (c_int, [POINTER(PatchEmbedDesc), c_int]),
    "me_patch_embed_wgrad_workspace_bytes": (c_size_t, [POINTER(PatchEmbedDesc), c_int, c_int]),
    "me_patch_embed_wgrad": (c_int, [POINTER(PatchEmbedDesc), c_void_p, c_int64, c_void_p, c_int, c_void_p, c_float, c_void_p]),
    "me_timeseries_embed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, POINTER(c_void_p), POINTER(c_int32),
                                    c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "me_timeseries_unfold": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "me_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                              c_float, c_int, c_float, c_void_p, c_void_p]),
    "me_grad_stats_workspace": (c_size_t, []),
    "me_grad_stats": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "me_adamw_prepare": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_void_p]),
    "me_adamw_step_segments": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_float, c_float, c_float,
                                       c_float, c_void_p, c_void_p, c_void_p]),
    "me_fps": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "me_knn": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "me_group_relative": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "me_pool_tokens": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "me_pool_tokens_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "me_resize_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "me_comm_unique_id": (c_int, [c_void_p]),
    "me_comm_init": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_int, c_int]),
    "me_comm_destroy": (c_int, [c_void_p]),
    "me_comm_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int64)]),
    "me_allreduce_bucket": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "me_comm_join": (c_int, [c_void_p, c_void_p]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libmetaenc.so and bind every declared symbol.  Raises MetaEncError if the library or a symbol is
    missing -- the product path never silently degrades to PyTorch ops."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise MetaEncError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m metatransformer_amd.build` "
            "(needs hipcc; cross-compiles gfx950 without a GPU). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MetaEncError(f"libmetaenc.so does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.me_abi_version() != 1:
        raise MetaEncError(f"libmetaenc.so ABI version {lib.me_abi_version()} != 1; rebuild it")
    # profiling scripts switch the weight-gradient side stream off from outside (tools/prof_round.sh: kernels that share the chip
    # have no duration of their own).  The library reads no environment: the host does, and uses the ABI call.
    if os.environ.get("ME_WGRAD_OVERLAP", "1") == "0":
        lib.me_block_bwd_overlap(0)
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().me_last_error()
        raise MetaEncError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def dtype_code(dt: torch.dtype, storage: bool = False) -> int:
    if dt == torch.float32:
        return ME_F32
    if dt == torch.bfloat16:
        return ME_BF16
    if storage and dt == torch.float16:
        return ME_F16
    raise MetaEncError(f"unsupported dtype {dt}: libmetaenc computes in float32 or bfloat16"
                       + ("" if storage else " (float16 tensors are converted with ops.cast at the boundary)"))


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream
