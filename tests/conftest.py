import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def lib_built():
    """Make sure libfishdx.so exists (hipcc cross-compiles without a GPU)."""
    from fish_diffusion_amd import _build, _lib
    if not os.path.exists(_lib.LIB_PATH):
        _build.build(verbose=False)
    return _lib.LIB_PATH
