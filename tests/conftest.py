import os
import sys

import pytest

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
