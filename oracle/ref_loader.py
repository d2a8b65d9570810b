"""Import the reference's OWN hot-path code, unmodified, from /root/reference
(TEST INFRASTRUCTURE -- see oracle/__init__.py).

Only usable where /root/reference exists (the build container).  Nothing under
tests -m gpu / smoke() / bench.py may call this; it exists to (a) validate
oracle/block_oracle.py and (b) generate tests/golden/ via oracle/make_golden.py.

Recipe: the in-tree timm-borrowed Block lives in
PointCloud/openpoints/models/layers/{attention,mlp,norm,activation,drop,helpers,weight_init}.py.
The package's __init__ pulls in CUDA extensions, so the seven files are loaded
one by one under a synthetic parent package that exposes exactly the names
attention.py:8-9 and mlp.py:7-8 import (`Mlp, DropPath, trunc_normal_, lecun_normal_,
create_norm, create_act`), plus a stub `easydict.EasyDict` (needed by norm.py:9).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("METAENC_REFERENCE_ROOT", "/root/reference")
_LAYERS = os.path.join(REF_ROOT, "PointCloud", "openpoints", "models", "layers")
_PKG = "_ref_openpoints_layers"


def reference_available() -> bool:
    return os.path.isfile(os.path.join(_LAYERS, "attention.py"))


def _stub_easydict():
    if "easydict" in sys.modules:
        return
    m = types.ModuleType("easydict")

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            self[k] = v

    m.EasyDict = EasyDict
    sys.modules["easydict"] = m


def _load(name: str):
    full = f"{_PKG}.{name}"
    if full in sys.modules:
        return sys.modules[full]
    spec = importlib.util.spec_from_file_location(full, os.path.join(_LAYERS, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[full] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_layers():
    """Returns the module object of the reference's attention.py (has .Block, .Attention)."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REF_ROOT}")
    _stub_easydict()
    if _PKG not in sys.modules:
        pkg = types.ModuleType(_PKG)
        pkg.__path__ = [_LAYERS]
        sys.modules[_PKG] = pkg
    pkg = sys.modules[_PKG]
    wi = _load("weight_init")
    pkg.trunc_normal_, pkg.lecun_normal_ = wi.trunc_normal_, wi.lecun_normal_
    _load("helpers")
    pkg.DropPath = _load("drop").DropPath
    pkg.create_norm = _load("norm").create_norm
    pkg.create_act = _load("activation").create_act
    pkg.Mlp = _load("mlp").Mlp
    return _load("attention")


def reference_encoder(depth: int, dim: int, num_heads: int, eps: float = 1e-5, mlp_ratio: float = 4.0):
    """nn.Sequential of the reference's in-tree Block, built as README.md:125-148 builds the
    timm one.  norm eps: README-style sites use nn.LayerNorm's default 1e-5; timm-factory
    sites use 1e-6 -- passed through norm_args."""
    import torch.nn as nn
    att = load_reference_layers()
    blocks = [att.Block(dim=dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=True,
                        norm_args={"norm": "ln", "eps": eps}, act_args={"act": "gelu"})
              for _ in range(depth)]
    return nn.Sequential(*blocks).eval()


def _load_file(modname: str, relpath: str, stubs: dict | None = None):
    """Load one reference file by path, with optional stub modules for un-installed imports."""
    for k, v in (stubs or {}).items():
        sys.modules.setdefault(k, v)
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF_ROOT, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def _timm_stub():
    """Minimal stand-in for the handful of timm helpers the reference tokenizer files import
    (to_2tuple / trunc_normal_ / drop_path / register_model) -- none of them is arithmetic
    on the tokenizer path."""
    import collections.abc
    from itertools import repeat

    def to_2tuple(x):
        if isinstance(x, collections.abc.Iterable) and not isinstance(x, str):
            return tuple(x)
        return tuple(repeat(x, 2))

    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    layers = types.ModuleType("timm.models.layers")
    registry = types.ModuleType("timm.models.registry")
    layers.to_2tuple = to_2tuple
    layers.trunc_normal_ = lambda t, std=.02, **kw: t.data.normal_(0, std)
    layers.drop_path = lambda x, p=0., training=False: x
    registry.register_model = lambda f: f
    timm.models, models.layers, models.registry = models, layers, registry
    return {"timm": timm, "timm.models": models, "timm.models.layers": layers,
            "timm.models.registry": registry}


def reference_image_patch_embed():
    return _load_file("_ref_d2s_image", "Data2Seq/Image.py").PatchEmbed


def reference_time_series_embedding():
    return _load_file("_ref_d2s_ts", "Data2Seq/Time_Series.py").DataEmbedding


def reference_acoustic_patch_embed():
    return _load_file("_ref_d2s_acoustic", "Data2Seq/Acoustic.py", _timm_stub()).PatchEmbed


def reference_video_module():
    return _load_file("_ref_video_ft", "Video/models/modeling_finetune.py", _timm_stub())


def reference_detection_vit_module():
    """Image/detection/mmdet_custom/models/backbones/base/vit.py, unmodified: WindowedAttention (pad / unfold / attend /
    fold / crop, :148-192), Block with layer_scale gamma1 / gamma2 (:276-335) and TIMMVisionTransformer.resize_pos_embed
    (:459-486).  The file imports mmcv / mmcv_custom / mmdet / mmengine / timm at module level, none installed here and
    none arithmetic on this path: BaseModule -> nn.Module, loggers / checkpoint loaders / initialisers -> no-ops.  The
    two timm layers it USES, Mlp and DropPath, are served from the reference's own in-tree twin of timm
    (PointCloud/openpoints/models/layers/{mlp,drop}.py) behind timm's 0.4.12 constructor signature."""
    import torch.nn as nn
    att = load_reference_layers()
    pkg = sys.modules[_PKG]

    class Mlp(pkg.Mlp):
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
            assert act_layer is nn.GELU
            super().__init__(in_features, hidden_features, out_features, act_args={"act": "gelu"}, drop=drop)

    stubs = dict(_timm_stub())
    # (an earlier loader call may already have installed the timm stub: _load_file keeps the first one)
    lay = sys.modules.get("timm.models.layers", stubs["timm.models.layers"])
    lay.Mlp = Mlp
    lay.DropPath = pkg.DropPath

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        return m
    noop = lambda *a, **k: None          # noqa: E731
    stubs.update({
        "mmcv": mod("mmcv"), "mmcv.runner": mod("mmcv.runner", BaseModule=nn.Module),
        "mmcv_custom": mod("mmcv_custom", my_load_checkpoint=noop),
        "mmdet": mod("mmdet"), "mmdet.utils": mod("mmdet.utils", get_root_logger=noop),
        "mmengine": mod("mmengine"), "mmengine.logging": mod("mmengine.logging", print_log=noop),
        "mmengine.model": mod("mmengine.model", BaseModule=nn.Module, ModuleList=nn.ModuleList),
        "mmengine.model.weight_init": mod("mmengine.model.weight_init", constant_init=noop, kaiming_init=noop, trunc_normal_=noop),
        "mmengine.runner": mod("mmengine.runner"),
        "mmengine.runner.checkpoint": mod("mmengine.runner.checkpoint", CheckpointLoader=object, load_state_dict=noop),
    })
    del att
    return _load_file("_ref_det_vit", "Image/detection/mmdet_custom/models/backbones/base/vit.py", stubs)


def reference_pointcloud_modules():
    """PointCloud/openpoints/models/layers/group_embed.py (PointPatchEmbed, :60-172) and
    models/classification/cls_base.py (ClsHead, :77-136), both UNMODIFIED, for the row-f4 fixtures.

    The files are loaded under a synthetic mirror of the package tree (`_ref_op.models.layers.*`,
    `_ref_op.models.classification.cls_base`) so that their relative imports resolve to the reference's own sibling files
    (conv.py / norm.py / activation.py / group.py / subsample.py / local_aggregation.py).  What cannot run here is stubbed,
    and only that:
      * `openpoints.cpp[.pointnet2_batch].pointnet2_cuda` (the built CUDA extension) and the two autograd Functions that
        call it: `furthest_point_sample` -> the literal CPU restatement of the kernel
        (oracle.tokenizer_oracle.fps_reference, itself compared on the GPU with the reference kernel built by
        oracle/build_ref.py), `grouping_operation` -> the gather it performs (group_points_gpu.cu:53-70: out[b,c,p,s] =
        features[b,c,idx[b,p,s]]);
      * the registry / checkpoint / loss helpers (`..build.MODELS`, `...utils`, `...loss`): decorators and loaders, no
        arithmetic.
    Returns (group_embed module, cls_base module)."""
    import torch
    import torch.nn as nn
    load_reference_layers()          # easydict stub (norm.py) etc.
    root = os.path.join(REF_ROOT, "PointCloud", "openpoints")

    def pkg(name, path=None, **attrs):
        m = sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            m.__path__ = [path] if path else []
            sys.modules[name] = m
        for k, v in attrs.items():
            setattr(m, k, v)
        return m

    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls

        def build(self, *a, **k):
            raise RuntimeError("registry stub: not used by the fixtures")

    noop = lambda *a, **k: None          # noqa: E731
    pkg("_ref_op", root)
    pkg("_ref_op.utils", get_missing_parameters_message=noop, get_unexpected_parameters_message=noop, load_checkpoint=noop,
        registry=types.SimpleNamespace(Registry=lambda *a, **k: _Registry()))
    pkg("_ref_op.loss", build_criterion_from_cfg=noop)
    models = pkg("_ref_op.models", os.path.join(root, "models"))
    pkg("_ref_op.models.build", MODELS=_Registry(), build_model_from_cfg=noop)
    layers = pkg("_ref_op.models.layers", _LAYERS)
    pkg("_ref_op.models.classification", os.path.join(root, "models", "classification"))
    ext = types.SimpleNamespace()        # the CUDA extension: never called (its two callers are replaced below)
    pkg("openpoints"); pkg("openpoints.cpp", pointnet2_cuda=ext); pkg("openpoints.cpp.pointnet2_batch", pointnet2_cuda=ext)

    def load(full, path):
        if full in sys.modules and getattr(sys.modules[full], "__file__", None):
            return sys.modules[full]
        spec = importlib.util.spec_from_file_location(full, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        spec.loader.exec_module(mod)
        return mod

    L = "_ref_op.models.layers."
    for name in ("weight_init", "helpers", "activation", "norm", "conv"):
        load(L + name, os.path.join(_LAYERS, name + ".py"))
    conv = sys.modules[L + "conv"]
    layers.create_linearblock = conv.create_linearblock
    sub = load(L + "subsample", os.path.join(_LAYERS, "subsample.py"))
    grp = load(L + "group", os.path.join(_LAYERS, "group.py"))

    def cpu_fps(xyz, npoint):
        from . import tokenizer_oracle as to
        return torch.from_numpy(to.fps_reference(xyz.detach().cpu().numpy(), int(npoint)))

    def cpu_grouping(features, idx):
        B, C, N = features.shape
        _, P, S = idx.shape
        return torch.gather(features.unsqueeze(2).expand(B, C, P, N), 3, idx.long().unsqueeze(1).expand(B, C, P, S))

    sub.furthest_point_sample = cpu_fps
    grp.grouping_operation = cpu_grouping
    load(L + "local_aggregation", os.path.join(_LAYERS, "local_aggregation.py"))
    ge = load(L + "group_embed", os.path.join(_LAYERS, "group_embed.py"))
    cb = load("_ref_op.models.classification.cls_base", os.path.join(root, "models", "classification", "cls_base.py"))
    del models, nn
    return ge, cb
