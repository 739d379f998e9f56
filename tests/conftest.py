import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as entry
    return entry.load_package()


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding.load()
