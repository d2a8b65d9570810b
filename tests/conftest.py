import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible (gpu-marked tests run on the MI355X box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    return torch.device("cuda:0")


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max-abs error normalised by the max-abs of the reference (the parity metric of SURVEY.md 8c)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# tolerances, stated once:
#   fp32 path  : 1e-3 relative (north star); in practice ~1e-6
#   bf16 path  : operands rounded to 8 mantissa bits, fp32 accumulation/statistics -> 3e-2 of max-abs per block stack
TOL_F32 = 1e-3
TOL_BF16 = 3e-2
