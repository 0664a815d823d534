import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True  # never leave .pyc files next to the (root-writable) reference mount


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# GPU suite order under `pytest -x` (the driver's invocation): kernels first, then the VAE, the denoiser, the BASELINE
# geometries, and the multi-process sequence-parallel cases LAST -- so that a failure in the most contention-sensitive
# tests cannot keep the per-kernel tests from running (round 2: one seqpar case cut 71 collected tests short).
_GPU_ORDER = ["test_gpu_kernels", "test_gpu_fp8", "test_gpu_rank_shapes", "test_gpu_stdit_shapes", "test_gpu_vae",
              "test_gpu_mmdit", "test_gpu_baseline_geometry", "test_gpu_overlap", "test_gpu_seqpar_1gpu", "test_gpu_seqpar_nccl"]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _GPU_ORDER.index(mod) if mod in _GPU_ORDER else -1   # CPU tests (and unknown modules) keep their place, first

    items.sort(key=rank)   # stable: the order inside a module is unchanged


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale) and load libosk_hip.so; GPU tests fail loudly when it cannot be loaded."""
    from open_sora_amd.build import build_lib

    build_lib()
    from open_sora_amd import _C

    return _C
