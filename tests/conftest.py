import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device; there is no CPU fallback")
    return torch.device("cuda:0")


def golden_weights(z):
    """{"w.<name>": float32 array, "wb.<name>": uint16 bf16 pattern} of a fixture -> {name: float32 array}."""
    import numpy as np

    out = {}
    for k in (z.files if hasattr(z, "files") else z):
        if k.startswith("w."):
            out[k[2:]] = z[k]
        elif k.startswith("wb."):
            out[k[3:]] = (z[k].astype(np.uint32) << 16).view(np.float32)
    return out
