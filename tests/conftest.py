import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_gpu_runtime_first():
    """torch carries its own copy of the HIP runtime.  When a process uses both that copy and the one
    libpbr355.so links, torch's must open the device first: the other order now and then ends in
    'No HIP GPUs are available' from torch.cuda (bench.py has always used this order)."""
    try:
        import torch
    except ImportError:
        return
    if torch.cuda.is_available():
        torch.cuda.init()


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as entry
    return entry.load_package()


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding.load()
