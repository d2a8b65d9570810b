"""GPU: the point-cloud half of SURVEY 8 row f4 pinned to the reference itself.

  * me_fps against the reference's OWN kernel -- PointCloud/openpoints/cpp/pointnet2_batch/src/sampling_gpu.cu:101-258,
    compiled unmodified for gfx950 by oracle/build_ref.py into oracle/_ref/libref_fps.so and run on the same clouds
    (ties, lattices, the block sizes opt_n_threads picks, the S3DIS cloud size);
  * PointPatchEmbed and ClsHead against tests/golden/pointcloud.npz, written by oracle/make_golden.py from the reference's
    own classes (group_embed.py:60-172, cls_base.py:77-136), parameters loaded with load_state_dict(strict=True).
"""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, TOL_F32, check_close, rel_err
from metatransformer_amd import heads

pytestmark = pytest.mark.gpu

REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_fps.so")
LAUNCHER = "_Z39furthest_point_sampling_kernel_launcheriiiPKfPfPi"       # sampling_gpu.cu:212


@pytest.fixture(scope="module")
def ref_fps():
    if not os.path.isfile(REF_SO):
        pytest.skip("oracle/_ref/libref_fps.so is not built (python -m oracle.build_ref in the build container; it travels "
                    "to the GPU box with the snapshot)")
    lib = ctypes.CDLL(REF_SO)
    fn = getattr(lib, LAUNCHER)
    fn.restype = None
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]

    def run(p: torch.Tensor, m: int) -> torch.Tensor:
        B, n, _ = p.shape
        idx = torch.zeros(B, m, dtype=torch.int32, device=p.device)
        temp = torch.full((B, n), 1e10, dtype=torch.float32, device=p.device)      # subsample.py:96 fill_(1e10)
        torch.cuda.synchronize()
        fn(B, n, m, p.data_ptr(), temp.data_ptr(), idx.data_ptr())                 # default stream, as the reference launches it
        torch.cuda.synchronize()
        return idx
    return run


def _clouds(n, seed):
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(4, n, 3, generator=g) * 2 - 1
    p[1, n // 2:] = p[1, : n - n // 2].clone()                   # exact duplicates: ties in every round
    p[2] = torch.round(p[2] * 4) / 4                             # a coarse lattice: many equal distances, exact arithmetic
    p[3] = p[3] * 1e3                                            # large coordinates: rounding of the distance decides
    return p


@pytest.mark.parametrize("n,m", [(2, 2), (5, 5), (33, 20), (64, 64), (100, 50), (777, 100), (1024, 256), (2048, 512), (3000, 64),
                                 (8192, 300), (24000, 1500)])
def test_fps_indices_equal_the_reference_kernel(dev, ref_fps, n, m):
    p = _clouds(n, 100 + n).to(dev).contiguous()
    want = ref_fps(p, m)
    got = heads.furthest_point_sample(p, m)
    assert got.dtype == torch.int32 and bool((got[:, 0] == 0).all())
    same = torch.equal(got, want)
    if not same:
        b, j = [int(v[0]) for v in torch.nonzero(got != want, as_tuple=True)]
        raise AssertionError(f"n={n} m={m}: first difference at cloud {b}, round {j}: me_fps {int(got[b, j])}, reference kernel {int(want[b, j])}")


def test_fps_memory_resident_form_equals_the_reference_kernel(dev, ref_fps):
    """n beyond the register-resident form (24 576 points): the same rounds with points and temp in memory"""
    p = _clouds(30000, 7)[:3].to(dev).contiguous()
    assert torch.equal(heads.furthest_point_sample(p, 200), ref_fps(p, 200))


def test_literal_restatement_equals_the_reference_kernel(dev, ref_fps):
    """oracle.tokenizer_oracle.fps_reference (used on CPU to generate tests/golden/pointcloud.npz) follows the same kernel"""
    from oracle import tokenizer_oracle as to
    for n, m in ((100, 40), (777, 60), (1500, 48)):
        p = _clouds(n, 300 + n)
        assert np.array_equal(to.fps_reference(p.numpy(), m), ref_fps(p.to(dev).contiguous(), m).cpu().numpy()), (n, m)


def _load(z, prefix):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def test_point_patch_embed_equals_the_reference_class(dev):
    z = np.load(os.path.join(GOLDEN, "pointcloud.npz"))
    cfg = json.loads(str(z["ppe/config"]))
    mod = heads.PointPatchEmbed(sample_ratio=cfg["sample_ratio"], group_size=cfg["group_size"], embed_dim=cfg["embed_dim"],
                                channels=tuple(cfg["channels"]), subsample=cfg["subsample"], group=cfg["group"],
                                feature_type=cfg["feature_type"], reduction=cfg["reduction"])
    mod.load_state_dict(_load(z, "ppe/w/"), strict=True)          # the reference module's own state dict
    mod = mod.to(dev).eval()
    p = torch.from_numpy(z["ppe/p"]).to(dev)
    (pp, center), (x, out_f) = mod(p)
    assert x is None and pp is p
    assert torch.equal(center.cpu(), torch.from_numpy(z["ppe/center"]))           # sampled centres: same points, same order
    check_close(out_f, torch.from_numpy(z["ppe/out_f"]), TOL_F32, "PointPatchEmbed out_f")


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_cls_head_equals_the_reference_class(dev, tag):
    z = np.load(os.path.join(GOLDEN, "pointcloud.npz"))
    kw = json.loads(str(z[f"cls_{tag}/config"]))
    head = heads.ClsHead(**kw)
    head.load_state_dict(_load(z, f"cls_{tag}/w/"), strict=True)
    head = head.to(dev).eval()
    x = torch.from_numpy(z[f"cls_{tag}/x"]).to(dev)
    check_close(head(x), torch.from_numpy(z[f"cls_{tag}/y"]), TOL_F32, f"ClsHead {tag} logits")
    if tag == "b":
        # the class defaults in TRAINING mode: nn.ReLU(inplace=True) directly behind the Linear (no norm layer in between)
        head.train()
        xr = x.clone().requires_grad_(True)
        y = head(xr)
        (y * torch.from_numpy(z["cls_b/go"]).to(dev)).sum().backward()
        check_close(y, torch.from_numpy(z["cls_b/y_train"]), TOL_F32, "ClsHead b train logits")
        check_close(xr.grad, torch.from_numpy(z["cls_b/dx"]), TOL_F32, "ClsHead b dx")
        for k, p in head.named_parameters():
            assert rel_err(p.grad, torch.from_numpy(z["cls_b/g/" + k])) < TOL_F32, k
