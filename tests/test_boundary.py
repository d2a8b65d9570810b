"""CPU: the drop-in boundary -- the C-ABI library loads and exports every symbol include/metaenc.h declares, the
ctypes mirror of the descriptor struct matches the C layout, the product package never touches oracle/, and the
host-side module mirrors the reference's plugin interface (names, shapes, error behaviour)."""
import ctypes
import os
import re
import subprocess
import tempfile

import pytest
import torch
import torch.nn as nn

from conftest import ROOT

import metatransformer_amd as M
from metatransformer_amd import _capi, build as me_build
from oracle import block_oracle as bo

HEADER = os.path.join(ROOT, "include", "metaenc.h")


@pytest.fixture(scope="module")
def lib():
    me_build.build(verbose=False)
    return _capi.load()


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(me_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_bound_and_exported(lib):
    names = header_functions()
    assert len(names) >= 18
    assert set(names) == set(_capi.SIGNATURES), "include/metaenc.h and _capi.SIGNATURES disagree"
    for n in names:
        assert hasattr(lib, n), f"libmetaenc.so does not export {n}"
    assert lib.me_abi_version() == 1
    assert lib.me_build_arch() == b"gfx950"


@pytest.mark.parametrize("cname,pyname", [("me_gemm_desc", "GemmDesc"), ("me_block_desc", "BlockDesc"),
                                          ("me_block_grads", "BlockGrads"), ("me_gemm_profile_rec", "GemmProfileRec"),
                                          ("me_adamw_segment", "AdamwSegment"), ("me_adamw_ctl", "AdamwCtl"),
                                          ("me_patch_embed_desc", "PatchEmbedDesc")])
def test_struct_layouts_match_c(cname, pyname):
    """every struct that crosses the C ABI: size and field offsets of the ctypes mirror == what gcc lays out from the header"""
    cls = getattr(_capi, pyname)
    fields = [n for n, _ in cls._fields_]
    code = ('#include <stdio.h>\n#include <stddef.h>\n#include "metaenc.h"\nint main(void){ printf("size %zu\\n", sizeof('
            + cname + '));\n' + "".join(f'printf("{f} %zu\\n", offsetof({cname}, {f}));\n' for f in fields) + "return 0; }\n")
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(code)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split("\n")
    got = dict(l.split() for l in out if l)
    assert int(got.pop("size")) == ctypes.sizeof(cls)
    for name in fields:
        assert int(got[name]) == getattr(cls, name).offset, name


def test_patch_embed_fuses_the_reference_geometries(lib):
    """host-side decision only (no kernel runs): the image and tubelet embeds of the reference gather inside the GEMM's operand stager
    when pixels and weight are bf16 -- since round 6 the spectrogram's stride-10 patches too (patch rows only have to be dword-aligned);
    fp32 arithmetic, odd patch shapes and odd image widths take the two-pass route and ask for a workspace"""
    def desc(shape, geom, xdt=_capi.ME_BF16, wdt=_capi.ME_BF16, cout=768):
        d = _capi.PatchEmbedDesc()
        d.x, d.x_dtype = 4096, xdt                                  # (an aligned address; never dereferenced by these two queries)
        B, Cin, T, H, W = shape
        d.B, d.Cin, d.T, d.H, d.W = B, Cin, T, H, W
        d.kt, d.kh, d.kw, d.st, d.sh, d.sw = geom
        d.w_dtype, d.Cout = wdt, cout
        return d
    image = desc((256, 3, 1, 224, 224), (1, 16, 16, 1, 16, 16))                      # Data2Seq/Image.py:19-28
    video = desc((8, 3, 16, 224, 224), (2, 16, 16, 2, 16, 16))                       # Video/models/modeling_finetune.py:283-297
    audio = desc((12, 1, 1, 128, 1024), (1, 16, 16, 1, 10, 10))                      # Audio/src/models/ast_models.py:86
    for d, want in ((image, 1), (video, 1), (audio, 1), (desc((2, 3, 1, 224, 224), (1, 16, 16, 1, 16, 16), xdt=_capi.ME_F32), 0),
                    (desc((2, 3, 1, 224, 224), (1, 16, 16, 1, 16, 16), xdt=_capi.ME_F32, wdt=_capi.ME_F32), 0),
                    (desc((2, 3, 1, 224, 224), (1, 14, 14, 1, 14, 14)), 0), (desc((2, 3, 1, 220, 220), (1, 16, 16, 1, 16, 16)), 1),
                    (desc((2, 3, 1, 224, 221), (1, 16, 16, 1, 16, 16)), 0), (desc((2, 1, 1, 128, 100), (1, 16, 16, 1, 10, 9)), 0),
                    (desc((2, 3, 1, 224, 224), (1, 16, 16, 1, 16, 16), cout=772), 0)):
        assert lib.me_patch_embed_fused(ctypes.byref(d)) == want
        assert (lib.me_patch_embed_workspace_bytes(ctypes.byref(d)) == 0) == bool(want)
    bad = desc((2, 3, 1, 8, 8), (1, 16, 16, 1, 16, 16))
    assert lib.me_patch_embed_fused(ctypes.byref(bad)) == 0 and lib.me_patch_embed(ctypes.byref(bad), None) != 0
    assert b"kernel larger than input" in lib.me_last_error()


def test_gemm_planning_queries_of_round_6(lib):
    """host-side decisions only (no kernel runs): me_gemm_takes_row_parts (LayerNorm partials consumed by the folded GEMM), me_gemm_takes_a_wrap
    (A as two planes with a wrapped reduction), and the workspace query under me_gemm_reserve_cus (weight gradients on fewer workgroups: the query
    must cover either plan)"""
    def desc(M_, N_, K_, op=_capi.ME_GEMM_NT, cdt=_capi.ME_BF16, lda=None, ldb=None):
        d = _capi.GemmDesc()
        d.op, d.ab_dtype, d.M, d.N, d.K = op, _capi.ME_BF16, M_, N_, K_
        d.A, d.lda = 4096, lda or (K_ if op == _capi.ME_GEMM_NT else M_)
        d.B, d.ldb = 8192, ldb or (K_ if op == _capi.ME_GEMM_NT else N_)
        d.C, d.ldc, d.c_dtype, d.alpha = 16384, N_, cdt, 1.0
        return d
    # row_parts: K = 256 .. 1024 in steps of 256, every CU a 256 x 256 tile, bf16 output, plain bias / GELU epilogue
    for M_, N_, K_, want in ((50432, 2304, 768, 1), (50432, 3072, 768, 1), (65536, 4096, 1024, 1), (4096, 2304, 768, 0), (50432, 2304, 384, 0),
                             (50432, 2304, 1280, 0)):
        d = desc(M_, N_, K_)
        d.col_shift, d.row_parts, d.row_nparts, d.row_eps = 32768, 65536, max(K_ // 256, 1), 1e-6
        assert lib.me_gemm_takes_row_parts(ctypes.byref(d)) == want, (M_, N_, K_)
    d = desc(50432, 2304, 768, cdt=_capi.ME_F32)
    d.col_shift, d.row_parts, d.row_nparts, d.row_eps = 32768, 65536, 3, 1e-6
    assert lib.me_gemm_takes_row_parts(ctypes.byref(d)) == 0              # (the resident kernel writes bf16)
    # a_wrap_k: the one-tile 256 x 256 family, plain or fp32-residual epilogue
    for M_, N_, Kc, want in ((50432, 768, 3072, 1), (8192, 1024, 4096, 1), (512, 768, 3072, 0)):
        d = desc(M_, N_, 3 * Kc, cdt=_capi.ME_F32, lda=2 * Kc)
        d.a_wrap_k = 2 * Kc
        assert lib.me_gemm_takes_a_wrap(ctypes.byref(d)) == want, (M_, N_, Kc)
    d = desc(50432, 768, 9216, cdt=_capi.ME_F32, lda=6144)
    d.a_wrap_k = 6144 + 64                                                 # not a multiple of 128
    assert lib.me_gemm_takes_a_wrap(ctypes.byref(d)) == 0
    # weight gradients: the workspace query with a reservation covers the default plan too (more slabs per tile, never fewer bytes)
    for Mo, No in ((2304, 768), (768, 3072), (768, 768)):
        d = desc(Mo, No, 50432, op=_capi.ME_GEMM_TN, cdt=_capi.ME_F32)
        w0 = lib.me_gemm_workspace_bytes(ctypes.byref(d))
        prev = lib.me_gemm_reserve_cus(16)
        try:
            w16 = lib.me_gemm_workspace_bytes(ctypes.byref(d))
        finally:
            assert lib.me_gemm_reserve_cus(prev) == 16
        assert w0 > 0 and w16 >= w0, (Mo, No, w0, w16)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "metatransformer_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "oracle/_ref" not in txt and "/root/reference" not in txt, f


def test_no_cpu_fallback():
    enc = M.build_encoder(1, 64, 2)
    with pytest.raises(M.MetaEncError, match="no CPU fallback"):
        enc(torch.randn(1, 3, 64))
    with pytest.raises(M.MetaEncError):
        M.PatchEmbed(img_size=32, embed_dim=64)(torch.randn(1, 3, 32, 32))
    with pytest.raises(M.MetaEncError):
        M.DataEmbedding(c_in=3, d_model=64).eval()(torch.randn(1, 8, 3))


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", "/nonexistent/libmetaenc.so")
    with pytest.raises(M.MetaEncError, match="not built"):
        _capi.load()


def test_block_mirrors_reference_interface():
    # README.md:125-135 construction, verbatim keyword set
    enc = nn.Sequential(*[M.Block(dim=64, num_heads=4, mlp_ratio=4., qkv_bias=True, norm_layer=nn.LayerNorm,
                                  act_layer=nn.GELU) for _ in range(3)])
    sd = bo.make_encoder_state_dict(3, 64, seed=0)
    assert list(enc.state_dict().keys()) == list(sd.keys())
    assert all(enc.state_dict()[k].shape == v.shape for k, v in sd.items())
    enc.load_state_dict(sd, strict=True)
    # strict=True must reject a foreign key set, like the reference call sites rely on
    bad = dict(sd); bad["0.attn.q_bias"] = torch.zeros(64)
    with pytest.raises(RuntimeError):
        enc.load_state_dict(bad, strict=True)
    # indexable / sliceable / iterable (vit_adapter.py:107, X-Ray/train.py:80)
    assert isinstance(enc[1], M.Block) and len(enc[0:2]) == 2 and len(list(enc)) == 3
    # freezing + optimizer grouping by name (Video/optim_factory.py:30-41,67-73)
    for p in enc.parameters():
        p.requires_grad = False
    assert not any(p.requires_grad for p in enc.parameters())
    names = [n for n, _ in enc.named_parameters()]
    assert "1.mlp.fc1.bias" in names and all(len(p.shape) == 1 or n.endswith(".weight") for n, p in enc.named_parameters())
    # eps routes: default 1e-5, timm-factory sites 1e-6 via partial
    from functools import partial
    assert M.Block(64, 4).eps == 1e-5
    assert M.Block(64, 4, norm_layer=partial(nn.LayerNorm, eps=1e-6)).eps == 1e-6
    assert M.Block(768, 32).attn.scale == 24 ** -0.5          # Graph call site: 32 heads, head_dim 24
    with pytest.raises(M.MetaEncError):
        M.Block(64, 5)


def test_set_fp32_mode_reaches_every_block_of_a_task_model():
    """one line at an fp32 call site selects the fp32-accurate three-product arithmetic for the whole encoder, wherever it sits"""
    import torch
    enc = M.build_encoder(3, 256, 4)
    model = torch.nn.ModuleDict({"tokenizer": torch.nn.Linear(7, 256), "encoder": enc, "head": torch.nn.Linear(256, 10)})
    assert all(b.fp32_mode == "3xbf16" for b in enc)              # the default since round 6 (Block.default_fp32_mode)
    assert M.set_fp32_mode(enc, "exact") == 3 and all(b.fp32_mode == "exact" for b in enc)
    assert M.set_fp32_mode(model, "3xbf16") == 3 and all(b.fp32_mode == "3xbf16" for b in enc)
    with pytest.raises(M.MetaEncError):
        M.set_fp32_mode(model, "tf32")


def test_tokenizer_parameter_names_match_reference():
    assert list(M.PatchEmbed().state_dict()) == ["proj.weight", "proj.bias"]
    assert M.PatchEmbed().proj.weight.shape == (768, 3, 16, 16)
    assert M.VideoPatchEmbed().proj.weight.shape == (768, 3, 2, 16, 16) and M.VideoPatchEmbed().num_patches == 1568
    assert M.AcousticPatchEmbed.num_tokens(128, 1024) == 1212       # Audio/src/models/ast_models.py:122
    ts = M.DataEmbedding(c_in=7, d_model=64)
    keys = list(ts.state_dict())
    assert "value_embedding.tokenConv.weight" in keys and "position_embedding.pe" in keys
    assert "temporal_embedding.hour_embed.emb.weight" in keys
    assert ts.value_embedding.tokenConv.weight.shape == (64, 7, 3)


def test_flop_model():
    # BASELINE.md section 3
    assert abs(M.encoder_flops_per_sample(197, 768, 12) / 1e9 - 34.895) < 0.01
    assert abs(M.encoder_flops_per_sample(512, 1024, 24) / 1e9 - 335.0) < 0.1


def test_resident_gemm_isa_has_no_inner_loop_spills_and_keeps_its_ticket_register():
    """tools/check_g3r_isa.py on the assembly hipcc emits for gemm3.hip: the hand-counted waits of the resident kernel assume
    that nothing the compiler does not show (a spill, a clobbered ticket register) sits inside its loops."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_g3r_isa", os.path.join(ROOT, "tools", "check_g3r_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with tempfile.TemporaryDirectory() as d:
        asm = mod.compile_asm(d)
    report, findings = mod.check(asm)
    assert len(report) >= 7 and not findings, findings


def test_attention_backward_isa_keeps_its_ticket_register():
    """The same check on the claimed-item draw of attn_bwd_ring16_kernel (attention.hip): the returning atomic is issued at the head
    of phase A and consumed at its end; nothing may write, spill or copy its result register in between."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_g3r_isa", os.path.join(ROOT, "tools", "check_g3r_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with tempfile.TemporaryDirectory() as d:
        asm = mod.compile_asm(d, mod.SRC_ATTN)
    report, findings = mod.check(asm, kernel="attn_bwd_ring16_kernel", check_loops=False)
    assert len(report) >= 8 and not findings, findings
    # the 16-wave kernels (ring and streaming forms) are built on four waves per SIMD: <= 128 registers, nothing spilled
    import re
    seen = 0
    for m in re.finditer(r"^(_Z\w*attn_(?:fwd|bwd)\w*(?:ring16|stream16)_kernel\w*):", asm, re.M):
        body = asm[m.start():asm.find(".end_amdhsa_kernel", m.start())]
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        # (the ring backward at head width 64 sits exactly at 128 registers and keeps one 8-byte value in scratch across an item)
        allowed = 4 if "attn_bwd_ring16" in m.group(1) else 0
        assert vgpr <= 128 and body.count("scratch_") <= allowed, (m.group(1), vgpr, body.count("scratch_"))
        seen += 1
    assert seen >= 10, seen      # ring16 forward / backward for every sub-tile count and both head widths, three streaming kernels


def test_fp32_block_that_cannot_run_three_products_says_so_once():
    """fp32_mode = "3xbf16" (default) covers the plain Block; the others run exact fp32 -- with ONE warning per reason, not silently"""
    import warnings
    import torch
    from metatransformer_amd import encoder

    class _B:
        fp32_mode, compute_dtype = "3xbf16", None

    encoder._exact_notes.discard("windowed attention")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        encoder._note_exact_fp32(_B(), torch.float32, torch.float32, "windowed attention")
        encoder._note_exact_fp32(_B(), torch.float32, torch.float32, "windowed attention")        # once
        encoder._note_exact_fp32(_B(), torch.bfloat16, torch.bfloat16, "windowed attention")      # not an fp32 block
        b = _B(); b.fp32_mode = "exact"
        encoder._exact_notes.discard("layer-scale gradient")
        encoder._note_exact_fp32(b, torch.float32, torch.float32, "layer-scale gradient")         # asked for exact: nothing to say
    assert len(w) == 1 and "exact-fp32 kernels" in str(w[0].message) and "windowed attention" in str(w[0].message)
