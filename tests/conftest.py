import os
import sys

import pytest

# torch BEFORE libpansim_hip.so: torch wheels bundle their own libamdhip64 / libhsa-runtime64 / librccl (same sonames as
# /opt/rocm's).  Imported first, they satisfy the library's dependencies and the process has ONE HIP runtime; imported after
# the library has pulled in /opt/rocm's, torch still loads its own copies by path and the process has two -- the second
# one to initialise may find no device ("No HIP GPUs are available" in a test that touches torch.cuda late in a session).
try:
    import torch  # noqa: F401
except ImportError:      # the product itself does not need torch (only pansim_amd.distributed does)
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (test infrastructure)"""
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def pa():
    """the product package; the HIP library must already be built (no fallback)"""
    import pansim_amd
    if not os.path.exists(pansim_amd.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    pansim_amd.load()
    return pansim_amd
