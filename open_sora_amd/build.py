"""Build libosk_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to the GPU box with the
source snapshot.  `python -m open_sora_amd.build` rebuilds unconditionally.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libosk_hip.so")
ARCH = "gfx950"


def sources() -> list[str]:
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale() -> bool:
    if not os.path.isfile(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG_DIR, "..", "include", "*.h"))
    return any(os.path.getmtime(s) > t for s in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libosk_hip.so (ROCm toolchain required)")
    os.makedirs(LIB_DIR, exist_ok=True)
    tmp = LIB_PATH + ".tmp"
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", tmp] + sources()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed ({r.returncode}):\n{r.stderr[-4000:]}")
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
