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


def pytest_sessionstart(session):
    """A fresh record of observed parity errors per session (tests/_parity.py)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _parity
    try:
        os.remove(_parity.log_path())
    except OSError:
        pass


def pytest_sessionfinish(session, exitstatus):
    import json
    import _parity
    summary = _parity.summarise()
    if summary:
        try:
            with open(os.path.splitext(_parity.log_path())[0] + ".json", "w") as f:
                json.dump({"rule": "abs 1e-4 where |want| <= 10, max(1e-4, rtol * |want|) beyond",
                           "tests": summary}, f, indent=1, sort_keys=True)
        except OSError:
            pass
