import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True  # never leave .pyc files next to the (root-writable) reference mount


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale) and load libosk_hip.so; GPU tests fail loudly when it cannot be loaded."""
    from open_sora_amd.build import build_lib

    build_lib()
    from open_sora_amd import _C

    return _C
