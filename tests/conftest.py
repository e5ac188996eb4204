import os
import sys

import pytest

# The oracle (oracle/liboracle.so) is OpenMP code, and so is part of torch.  On the GPU box the process sees every core of the host but may use a share of
# them (a CPU quota): a parallel region of one thread per VISIBLE core then runs into the quota and every oracle call costs a scheduler period (measured:
# 0.12 s per call whatever its size; the GPU suite took 550 s with it and takes 40 s without).  Eight threads are plenty for test-sized inputs.  Set before
# anything loads an OpenMP runtime; an explicit OMP_NUM_THREADS in the environment wins.  (oracle/pyoracle.py also sets the count through the runtime's API.)
os.environ.setdefault("OMP_NUM_THREADS", "8")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_report_header(config):
    """The library has a few measurement switches read from the environment (HEVCASM_HIP_LIB: another build of the library;
    HEVCASM_SAD4_NOWINDOW / _GENERIC, HEVCASM_PRED_NOFAST: other kernel paths).  A stale one in the shell would silently change
    what is tested: say so at the top of every run."""
    stale = sorted(k for k in os.environ if k.startswith("HEVCASM_"))
    from hevcasm_amd import capi
    lines = ["hevcasm library under test: %s" % capi.LIB_PATH]
    if stale:
        lines.append("WARNING: %s set in the environment -- this run does NOT test the default configuration" % ", ".join("%s=%s" % (k, os.environ[k]) for k in stale))
    return lines


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
