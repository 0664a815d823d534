"""GPU: kernels of this library must repeat bit for bit while OTHER kernels of this library run on a second stream of the same
process -- the production condition of the sequence-parallel step (collective / copy kernels and the next block's GEMMs
co-resident with compute) and of two ranks sharing a GPU.

Round 3 finding behind this file (profiles/r03_cross_kernel_interference.md): hipcc's packed-FP32 sequences (v_pk_fma_f32 &
co.) returned a wrong low half now and then while a workgroup of the MFMA kernels was resident on the same CU; the batched
adaLN GEMV was the victim that turned GPUTEST_r02 red.  The library is now built with -packed-fp32-ops (open_sora_amd/build.py);
these tests keep every compiler-scheduled kernel of the denoise step under that co-residency."""
import pytest
import torch

gpu = pytest.mark.gpu

VICTIMS = ["gemv", "ln", "qknorm", "gemm_small", "gemm256p", "attn"]
AGGRESSORS = ["gemm256p", "attn"]


@gpu
@pytest.mark.parametrize("aggressor", AGGRESSORS)
@pytest.mark.parametrize("victim", VICTIMS)
def test_kernel_repeats_while_mfma_kernel_runs_on_second_stream(hip_lib, victim, aggressor):
    from tools.xproc_probe import make_kernel

    side = torch.cuda.Stream()
    launch, outputs, keep = make_kernel(victim)
    with torch.cuda.stream(side):
        alaunch, _, akeep = make_kernel(aggressor)
    launch()
    torch.cuda.synchronize()
    first = [t.clone() for t in outputs()]
    iters = 600
    flags = torch.zeros(iters, dtype=torch.bool, device="cuda")
    for i in range(iters):
        with torch.cuda.stream(side):
            for _ in range(6):
                alaunch()
        for t in outputs():
            t.zero_()
        launch()
        f = torch.zeros((), dtype=torch.bool, device="cuda")
        for a, b in zip(outputs(), first):
            f = f | (a != b).any()
        flags[i] = f
    torch.cuda.synchronize()
    bad = int(flags.sum())
    assert bad == 0, f"{victim}: {bad} of {iters} launches differ from the first while {aggressor} runs on a second stream"


def test_library_has_no_packed_fp32_instructions(hip_lib):
    """the build flag that removes the victim pattern is in force: no v_pk_*_f32 / v_pk_mov_b32 in the shipped code object"""
    import os
    import shutil
    import subprocess

    from open_sora_amd.build import LIB_PATH

    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("ROCm LLVM tools not found")
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        lib = shutil.copy(LIB_PATH, os.path.join(d, "libosk_hip.so"))    # --offloading extracts NEXT TO the input file
        r = subprocess.run([objdump, "--offloading", lib], capture_output=True, text=True, cwd=d)
        cos = sorted(os.path.join(d, f) for f in os.listdir(d) if "gfx950" in f)   # one code object per translation unit
        if not cos:
            pytest.skip("could not extract the gfx950 code objects: " + r.stderr[-200:])
        dis = "".join(subprocess.run([objdump, "-d", c], capture_output=True, text=True).stdout for c in cos)
    assert dis.count("s_endpgm") > 50, "disassembly looks empty"
    hits = [ln for ln in dis.splitlines() if "v_pk_fma_f32" in ln or "v_pk_mul_f32" in ln or "v_pk_add_f32" in ln or "v_pk_mov_b32" in ln]
    assert not hits, hits[:5]
