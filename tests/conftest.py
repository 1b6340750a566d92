import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("rust-pathtracer_amd")


@pytest.fixture(scope="session")
def oracle(pkg):
    """The CPU oracle (oracle/libptref.so), built on demand.  Test infrastructure only."""
    import oracle_loader
    return oracle_loader.load(pkg)


@pytest.fixture(scope="session")
def engine(pkg):
    """The product: the HIP engine behind the C ABI.  No fallback."""
    return pkg.load()
