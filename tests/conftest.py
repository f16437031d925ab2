import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/memc_oracle.c), compiled on demand with gcc."""
    from oracle import memc_oracle
    memc_oracle.build()
    memc_oracle.lib()
    return memc_oracle


@pytest.fixture(scope="session")
def hip_lib_path():
    """Path of the in-tree libmemc_hip.so, built on demand (hipcc cross-compiles without a GPU)."""
    path = os.path.join(ROOT, "memc-net_amd", "lib", "libmemc_hip.so")
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    return path
