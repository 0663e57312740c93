import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """Fails loudly (never skips) when the GPU or the HIP library is missing."""
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    import distributions_amd as da

    da.lib()  # raises if libm355dist.so is missing
    return torch.device("cuda:0")
