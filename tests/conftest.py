import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; oracle/lpx_oracle.h)."""
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def lpx():
    """The product package; requires liblpx.so built in-tree (no CPU fallback)."""
    import linear_programming_solver_lpr381_amd as L
    L._lib.lib()
    return L


@pytest.fixture(scope="session")
def gpu(lpx):
    if lpx._lib.lib().lpx_device_count() < 1:
        pytest.fail("gpu-marked test running without a visible HIP device")
    lpx._lib.check(lpx._lib.lib().lpx_init(0))
    return lpx
