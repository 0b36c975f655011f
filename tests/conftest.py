import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name has a hyphen, so import it by string)."""
    return importlib.import_module("mpc-sensorlessao_amd")


@pytest.fixture(scope="session")
def gpu(pkg):
    """GPU tests must run the HIP path: no device -> fail loudly, never skip or fall back."""
    import torch
    assert torch.cuda.is_available(), "GPU test started without a visible HIP device"
    pkg.load()
    return torch.device("cuda:0")
