import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def refproc():
    """The reference's own C path (oracle/_ref/refdrv).  Present wherever `make -C oracle` ran with
    /root/reference available; the prebuilt binary also travels to the GPU box."""
    from oracle import pyref
    if not pyref.available():
        pytest.skip("oracle/_ref/refdrv not built")
    p = pyref.RefProc()
    yield p
    p.close()


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle
