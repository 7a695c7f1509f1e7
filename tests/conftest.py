import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rel_l2(a, b):
    a = torch.as_tensor(a)
    b = torch.as_tensor(b)
    if a.is_complex() or b.is_complex():
        a = a.to(torch.complex128)
        b = b.to(torch.complex128)
    else:
        a = a.to(torch.float64)
        b = b.to(torch.float64)
    denom = torch.linalg.norm(b.reshape(-1))
    num = torch.linalg.norm((a.cpu() - b.cpu()).reshape(-1))
    return (num / denom).item() if denom > 0 else num.item()


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _default_dtype_is_float32():
    """Several tests switch the torch default dtype (the reference's drivers do): none may leak into the next."""
    torch.set_default_dtype(torch.float32)
    yield
    torch.set_default_dtype(torch.float32)
