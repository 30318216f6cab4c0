import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The C oracle (oracle/libray_oracle.so), built on demand.  Test infrastructure only."""
    import orc as _orc
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def ctx():
    import ray_amd
    return ray_amd.default_context()


def golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
