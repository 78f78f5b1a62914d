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
    return importlib.import_module("object-oriented-slam_amd")


@pytest.fixture(scope="session")
def po():
    """C oracle binding (test infrastructure); builds oracle/libesl_oracle.so if needed."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def ctx(pkg):
    """A HIP context on cuda:0 through the C-ABI.  Fails loudly (no fallback) without a device."""
    c = pkg.Context(0)
    yield c
    c.close()
