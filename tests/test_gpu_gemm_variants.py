"""The opt-in / A/B instantiations of the large-tile GEMM run the SAME parity cases as the default one.  The switches are read
once per process (static initialisers in csrc/gemm_bf16.hip), so each variant gets its own pytest process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = {
    "4-wave persistent on 32x32x16 MFMAs (gemm256w.hip)": {"OSK_GEMM_X": "0"},
    "8-wave persistent, schedule 2 (gemm256p.hip)": {"OSK_GEMM_W4": "0"},
    "round-1 one-tile-per-workgroup kernel (gemm256.hip)": {"OSK_GEMM_PERSIST": "0"},
}


@pytest.mark.parametrize("name", list(VARIANTS))
def test_gemm_variant_passes_the_gemm_parity_cases(hip_lib, name):
    env = dict(os.environ, **VARIANTS[name])
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_kernels.py"), "-q", "-x", "-m", "gpu",
                        "-k", "gemm and not fp8", "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, f"{name}:\n{tail}\n{r.stderr[-2000:]}"
    assert " passed" in tail and "failed" not in tail, tail
