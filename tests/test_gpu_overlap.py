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


@gpu
@pytest.mark.parametrize("aggressor", AGGRESSORS)
def test_foreign_torch_elementwise_victim_under_mfma_aggressor(hip_lib, aggressor):
    """kernels of OTHER libraries are still compiled with packed FP32.  A torch f32 elementwise kernel
    (addcmul: a fused multiply-add per element, the v_pk_fma_f32 pattern) runs on the main stream while one of this library's
    MFMA kernels runs on a second stream.  This library cannot fix a foreign kernel: a mismatch here is the documented
    co-residency constraint observed, reported as xfail with its count -- not a defect of the
    product path, whose own kernels are covered by the test above."""
    from tools.xproc_probe import make_kernel

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        alaunch, _, akeep = make_kernel(aggressor)
    g = torch.Generator(device="cuda").manual_seed(3)
    a, b, c = (torch.randn(1 << 22, device="cuda", generator=g) for _ in range(3))
    first = torch.addcmul(c, a, b, value=1.5)
    torch.cuda.synchronize()
    iters = 400
    flags = torch.zeros(iters, dtype=torch.int32, device="cuda")
    for i in range(iters):
        with torch.cuda.stream(side):
            for _ in range(6):
                alaunch()
        out = torch.addcmul(c, a, b, value=1.5)
        flags[i] = (out != first).sum()
    torch.cuda.synchronize()
    bad = int((flags > 0).sum())
    if bad:
        pytest.xfail(f"torch.addcmul (foreign, packed-FP32 build): {bad} of {iters} launches differ, {int(flags.sum())} elements, while "
                     f"{aggressor} runs on a second stream -- the co-residency constraint of INTEGRATION.md section 4")


def test_library_has_no_packed_fp32_instructions(hip_lib):
    """the build flag that removes the victim pattern is in force: no v_pk_*_f32 / v_pk_mov_b32 in the shipped code objects"""
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
    import re

    hits, fn = [], ""
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", ln)
        if m:
            fn = m.group(1)
        elif "v_pk_fma_f32" in ln or "v_pk_mul_f32" in ln or "v_pk_add_f32" in ln or "v_pk_mov_b32" in ln:
            hits.append((fn[:60], ln.strip()[:80]))
    assert not hits, hits[:5]


# ------------------------------------------------------------------------------------------------------------------------
# the sequence-parallel step with P ranks as threads of this process on the one GPU (tests/local_transport.py): every rank has
# its own compute stream and its own communication stream, the K / V^T exchange really overlaps the Q / MLP-up GEMMs
SP_CASES = [
    (2, "hd72_eager_split", (2, 4, 8, 8, 64)),     # L = 320, 160 per rank: whole + ragged 64-key tiles per segment
    (4, "hd72_eager_split", (1, 4, 8, 8, 64)),     # 4 ranks: 8 heads / 4, 80 tokens per rank
    (2, "hd128_liger_split", (3, 2, 9, 7, 22)),    # L = 148, CFG-triple batch, liger RoPE
]


@gpu
@pytest.mark.parametrize("mode", ["allgather", "ulysses"])
@pytest.mark.parametrize("case", SP_CASES, ids=lambda c: f"w{c[0]}-{c[1]}")
def test_seqpar_ranks_as_threads_overlapped_exchange_equals_serial_order(hip_lib, case, mode):
    """north_star: the exchange around attention runs on a second HIP stream, overlapped with compute.  No multi-GPU node is
    available to this suite, so the overlap is exercised here: device-copy collectives on per-rank communication streams
    (same event protocol as DistTransport), P x 2 streams live on one GPU.  Overlapped == serial order bit for bit, every
    rank returns the same prediction, three forwards repeat, and the result matches the single-GPU forward and the oracle."""
    import copy

    from open_sora_amd import mmdit, seqpar
    from oracle import configs, mmdit_oracle as O
    from tests.local_transport import LocalTransport, run_ranks
    from tests.util import rel_l2, torch_inputs, torch_params

    P, name, geom = case
    cfg = configs.GOLDEN[name][0]
    if mode == "ulysses" and cfg["num_heads"] % P:
        pytest.skip("head exchange needs num_heads % P == 0")
    B, T, h, w, L_txt = geom
    dev = "cuda:0"
    model = mmdit.Flux(device_map=dev, torch_dtype=torch.bfloat16, **cfg)
    model.load_state_dict(torch_params(cfg, dtype=torch.bfloat16, device=dev), strict=True)
    inp = torch_inputs(cfg, B, T, h, w, L_txt, dtype=torch.bfloat16, device=dev)
    with torch.inference_mode():
        single = model(**inp).float().cpu()          # also builds the (shared, read-only) plan before the threads start
    torch.cuda.synchronize()

    def run(overlap):
        def rank_fn(rank, world):
            m = copy.copy(model)                       # shares parameters and plan; own sequence-parallel state and workspaces
            m.forward = m.forward_ckpt
            object.__setattr__(m, "_osk_ws_cache", {})
            tp = LocalTransport(world, rank, dev, overlap=overlap)
            sp = seqpar.enable(m, mode=mode, transport=tp)
            assert m._sp is sp and sp.P == P and sp.rank == rank
            with torch.inference_mode():
                outs = [m(**inp).float() for _ in range(3)]
            torch.cuda.current_stream().synchronize()
            return [o.cpu() for o in outs], tp.calls

        return run_ranks(P, rank_fn, dev)

    serial, over = run(False), run(True)
    n_blocks = cfg["depth"] + cfg["depth_single_blocks"]
    for res in (serial, over):
        for outs, calls in res:
            assert calls >= 3 * (2 * n_blocks + 1)     # K and V (or q, k, v, o) per block + the output gather, per forward
            assert all(torch.equal(outs[0], o) for o in outs[1:]), "sequence-parallel forward is not repeatable"
            assert torch.equal(outs[0], res[0][0][0]), "ranks disagree on the gathered prediction"
    assert torch.equal(over[0][0][0], serial[0][0][0]), "overlapped exchange differs from the serial order"
    with torch.inference_mode():
        truth = O.forward(torch_params(cfg), cfg, **torch_inputs(cfg, B, T, h, w, L_txt))
        ref_bf16 = O.forward(torch_params(cfg, dtype=torch.bfloat16), cfg, **torch_inputs(cfg, B, T, h, w, L_txt, dtype=torch.bfloat16))
    e_ref, e_sp = rel_l2(ref_bf16.float(), truth), rel_l2(over[0][0][0], truth)
    assert e_sp <= max(1.5 * e_ref, 2.0 ** -8), (e_sp, e_ref)
    assert rel_l2(over[0][0][0], single) <= 2.0 ** -7


@gpu
@pytest.mark.parametrize("P,mode", [(2, "allgather"), (4, "allgather"), (4, "ulysses")])
def test_seqpar_ranks_as_threads_at_the_bench_length(hip_lib, P, mode):
    """sequence parallelism had only been exercised at L <= 320.  Here at the BENCH token count
    (MMDiT-XL width, L = 16,384 + 512, B = 1; depth 2 + 4 so that the test stays short -- the exchange protocol does not depend
    on the depth) with P ranks as threads of this process on the one GPU: every rank's attention call has Lq = L / P >= 4,224
    rows against P key segments of 66 / 132 tiles, i.e. the wide FAST body with loader events at every segment end, beside the
    other ranks' GEMMs and copy "collectives" on 2 P streams.  Overlapped exchange == serial order bit for bit, all ranks agree,
    repeatable, and equal to the single-GPU forward at the kernels' own rounding level.  The per-forward times (serial vs
    overlapped, all ranks sharing ONE GPU: not a scaling measurement) go to gpurun_out/ for profiles/."""
    import copy
    import json
    import os
    import time

    from open_sora_amd import configs as pcfg, mmdit, seqpar
    from oracle import synth
    from tests.local_transport import LocalTransport, run_ranks
    from tests.util import fast_params, rel_l2

    cfg = dict(pcfg.MMDIT["XL"], depth=2, depth_single_blocks=4)
    dev = "cuda:0"
    model = mmdit.Flux(device_map=dev, torch_dtype=torch.bfloat16, **cfg)
    model.load_state_dict({k: v.to(dev, torch.bfloat16) for k, v in fast_params(synth.mmdit_param_shapes(cfg), seed=5).items()}, strict=True)
    inp = {k: torch.from_numpy(v) for k, v in synth.mmdit_inputs(cfg, 1, 16, 32, 32, 512).items()}
    inp = {k: (v.to(dev) if "ids" in k else v.to(dev, torch.bfloat16)) for k, v in inp.items()}
    with torch.inference_mode():
        single = model(**inp).float().cpu()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            model(**inp)
        torch.cuda.synchronize()
        t_single = (time.perf_counter() - t0) / 3 * 1e3
    assert model.attention_report(1, 16896)["bodies"] == ["attn_asm72_kernel<FAST>"]

    def run(overlap):
        def rank_fn(rank, world):
            m = copy.copy(model)
            m.forward = m.forward_ckpt
            object.__setattr__(m, "_osk_ws_cache", {})
            tp = LocalTransport(world, rank, dev, overlap=overlap)
            sp = seqpar.enable(m, mode=mode, transport=tp)
            with torch.inference_mode():
                outs = [m(**inp).float()]                               # warm-up: workspaces, exchange buffers
                torch.cuda.current_stream().synchronize()
                world.barrier.wait()
                t1 = time.perf_counter()
                outs += [m(**inp).float() for _ in range(2)]
                torch.cuda.current_stream().synchronize()
                world.barrier.wait()
                dt = (time.perf_counter() - t1) / 2 * 1e3
            return [o.cpu() for o in outs], dt, sp.head_parallel(cfg["num_heads"])

        return run_ranks(P, rank_fn, dev)

    serial, over = run(False), run(True)
    for res in (serial, over):
        for outs, _, heads in res:
            assert heads == (mode == "ulysses")
            assert all(torch.equal(outs[0], o) for o in outs[1:]), "sequence-parallel forward is not repeatable"
            assert torch.equal(outs[0], res[0][0][0]), "ranks disagree on the gathered prediction"
    assert torch.equal(over[0][0][0], serial[0][0][0]), "overlapped exchange differs from the serial order"
    assert rel_l2(over[0][0][0], single) <= 2.0 ** -7
    rec = dict(test="seqpar ranks-as-threads, XL width depth 2+4, L=16896, B=1, ONE GPU shared by all ranks", P=P, mode=mode,
               single_gpu_forward_ms=round(t_single, 2), serial_exchange_ms=round(max(r[1] for r in serial), 2),
               overlapped_exchange_ms=round(max(r[1] for r in over), 2))
    print(json.dumps(rec))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "sp_onegpu_bench_length.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
