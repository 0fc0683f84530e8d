import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """Builds (if stale) and loads libex4d_hip.so; GPU tests fail loudly if that is impossible."""
    from ex4dgs_amd import build, _C
    build.build()
    return _C.load()
