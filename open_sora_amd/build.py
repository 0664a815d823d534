"""Build libosk_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to the GPU box with the
source snapshot.  `python -m open_sora_amd.build` rebuilds unconditionally.

Every .hip file is its own translation unit: they are compiled to objects in parallel (one hipcc process per
file, object files cached under lib/obj/ and re-used while neither the source nor anything it can include
changed) and linked into the shared library.
"""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libosk_hip.so")
ARCH = "gfx950"
# -packed-fp32-ops: hipcc must not emit v_pk_{fma,mul,add}_f32 / v_pk_mov_b32 in the code it schedules.  Measured on MI355X
# (round 3, tools/interfere_probe.py + tools/xproc_probe.py, profiles/r03_cross_kernel_interference.md): a wave running such
# packed-FP32 sequences returns a wrong LOW half now and then while a workgroup of one of this library's MFMA kernels
# (gemm256p, attn_asm72) is resident on the same CU -- from another stream or another process.  That made the batched adaLN
# GEMV of one rank non-repeatable when two sequence-parallel ranks shared a GPU (GPUTEST_r02).  The same victim kernels built
# without the feature never mismatch.  The hand-written loops (generated .inc bodies) contain no packed-FP32 instruction.
# The feature string is a device-side one; hipcc hands -Xclang options to the x86 host pass too, which ignores it with a warning.
# (ADVICE r3 asked for -Xarch_device: this clang rejects "-Xarch_device -Xclang ..." -- "options requiring arguments are unsupported"
# -- and the argument-free spellings -Xarch_device -mno-packed-fp32-ops / -mattr=-packed-fp32-ops are accepted and do NOTHING:
# v_pk_fma_f32 is still emitted.  tests/test_gpu_overlap.py::test_library_has_no_packed_fp32_instructions disassembles the result.)
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# (Round 6 measured an exception for gemm256x.hip -- one wave per SIMD owns the whole register file, so its own packed code can never meet
# the victim condition: the compiler's packing there is worth 0.06 - 0.31 ms of 40.6 ms of GEMM time per step, a hand-written packed GELU
# nothing once the wait states behind its transcendentals were in.  Not adopted: one rule for every file.  profiles/r06g_*.)
# the flags are part of what the library IS (a build without -packed-fp32-ops is a wrong build, see above): their hash is stored
# next to the .so and a library built with other flags -- or by an A/B script of tools/ into this path -- is stale
STAMP_PATH = LIB_PATH + ".flags"


def flags_for(src: str) -> list[str]:
    """compile flags of one translation unit (the same for all of them)"""
    return FLAGS


def _flags_stamp() -> str:
    return hashlib.sha256(" ".join(FLAGS).encode()).hexdigest()[:16]


def sources() -> list[str]:
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _shared_deps() -> list[str]:
    """files any translation unit may include: the headers and the generators' output (*.inc: where every hot loop
    lives), plus the generators themselves (a changed generator whose output was not re-emitted is caught by
    tests/test_generated_kernels.py; here it only forces a rebuild)"""
    return (glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) +
            glob.glob(os.path.join(PKG_DIR, "..", "include", "*.h")) +
            glob.glob(os.path.join(PKG_DIR, "..", "tools", "gen_*_asm.py")))


def _newest(paths) -> float:
    return max((os.path.getmtime(p) for p in paths), default=0.0)


def _stale() -> bool:
    if not os.path.isfile(LIB_PATH):
        return True
    try:
        if open(STAMP_PATH).read().strip() != _flags_stamp():
            return True
    except OSError:
        return True
    return _newest(sources() + _shared_deps() + [os.path.abspath(__file__)]) > os.path.getmtime(LIB_PATH)


def _hipcc() -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libosk_hip.so (ROCm toolchain required)")
    return hipcc


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB_PATH
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    try:
        same_flags = open(STAMP_PATH).read().strip() == _flags_stamp()
    except OSError:
        same_flags = False
    force = force or not same_flags   # cached objects were compiled with other flags
    shared_t = _newest(_shared_deps() + [os.path.abspath(__file__)])

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        if not force and os.path.isfile(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), shared_t):
            return obj
        cmd = [hipcc, *flags_for(src), "-c", src, "-o", obj + ".tmp"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {os.path.basename(src)} ({r.returncode}):\n{r.stderr[-4000:]}")
        os.replace(obj + ".tmp", obj)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    for stale_obj in set(glob.glob(os.path.join(OBJ_DIR, "*.o"))) - set(objs):   # objects of deleted translation units
        os.remove(stale_obj)
    tmp = LIB_PATH + ".tmp"
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc link failed ({r.returncode}):\n{r.stderr[-4000:]}")
    os.replace(tmp, LIB_PATH)
    with open(STAMP_PATH, "w") as f:
        f.write(_flags_stamp() + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force="--incremental" not in sys.argv, verbose=True))
