"""The hand-scheduled K loops are GENERATED (tools/gen_attn_asm.py, tools/gen_gemm_asm.py, tools/gen_conv_sw_asm.py -> csrc/*.inc): the committed
.inc files must be exactly what the committed generators emit, and the generators' own bookkeeping must hold."""
import filecmp
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "open_sora_amd", "csrc")
TOOLS = os.path.join(ROOT, "tools")


def _regen(script, tmp_path, extra=()):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), "--out", str(tmp_path), *extra], check=True)
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(str(tmp_path), "*.inc")))


def test_attention_bodies_are_reproducible(tmp_path):
    names = _regen("gen_attn_asm.py", tmp_path)
    assert names, "generator wrote nothing"
    for n in names:
        assert filecmp.cmp(os.path.join(str(tmp_path), n), os.path.join(CSRC, n), shallow=False), f"{n} is stale: re-run tools/gen_attn_asm.py"


def test_gemm_and_conv_bodies_are_reproducible(tmp_path):
    names = _regen("gen_gemm_asm.py", tmp_path)
    assert names
    for n in names:
        assert filecmp.cmp(os.path.join(str(tmp_path), n), os.path.join(CSRC, n), shallow=False), f"{n} is stale: re-run tools/gen_gemm_asm.py"


def test_sliding_window_conv_bodies_are_reproducible(tmp_path):
    names = _regen("gen_conv_sw_asm.py", tmp_path)
    assert names
    for n in names:
        assert filecmp.cmp(os.path.join(str(tmp_path), n), os.path.join(CSRC, n), shallow=False), f"{n} is stale: re-run tools/gen_conv_sw_asm.py"


def test_wide_attention_schedule_invariants():
    """the wide head_dim-72 layout (4 query blocks per wave, one 32-key half per body): MFMA counts of prologue + 4 bodies + exit,
    fragment reads per body (5 K + 5 V^T: half of the 2-block layout's), register ranges, one event check per advance site"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_attn_asm as G

    L = G.LayoutW()
    st = G.generate_wide(L)
    text = "\n".join(st.lines)
    loop, events = text.split("\n.L@@_ev", 1)
    assert text.count("v_mfma_f32_32x32x16_bf16") == 20 + 4 * 20 and text.count("v_mfma_f32_16x16x32_bf16") == 4 * 40 + 8
    assert "v_max" not in text and "v_cndmask" not in loop and "ds_write_b32" not in loop
    assert loop.count("s_cbranch_scc1 .L@@_ev") == 3 + 4           # prologue K, V, K + one loader per body
    bodies = re.split(r"\n\.L@@_body\d\d:", loop)[1:]
    assert len(bodies) == 4
    for b in bodies:
        assert b.count("ds_read_b128") == 10 and b.count("s_barrier") == 1
        assert b.count("v_exp_f32") == 64 and b.count("v_cvt_pk_bf16_f32") == 32 and b.count("v_permlane16_swap_b32") == 16
        assert b.count("global_load_lds_dwordx4") == 3
    for m in re.finditer(r"\bv(\d+)\b", text):
        assert L.V_FIRST <= int(m.group(1)) < L.V_END, m.group(0)
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        assert L.V_FIRST <= int(m.group(1)) and int(m.group(2)) < L.V_END, m.group(0)
    for m in re.finditer(r"\ba\[(\d+):(\d+)\]", text):
        assert int(m.group(2)) < L.A_END
    for m in re.finditer(r"\bs(\d+)\b", text):
        assert G.S_FIRST <= int(m.group(1)) <= G.S_LAST, m.group(0)
    assert L.V_END <= 256 and L.A_END <= 256 and len(L.OPERANDS) <= 30
    # every O^T accumulator quad is written by exactly 2 x (number of bodies) + ... MFMAs: all 40 quads appear in every body
    quads = set(re.findall(r"v_mfma_f32_16x16x32_bf16 a\[(\d+):\d+\]", bodies[0]))
    assert len(quads) == 40


def test_generated_loops_fit_the_instruction_cache():
    """the hot loop of every hand-scheduled kernel has to stay inside the 64 KB instruction cache (shared by
    two CUs) with room to spare.  Measured on the BUILT library: per kernel of the disassembled gfx950 code objects, the longest
    backward branch (target .. branch) = the largest loop.  Budget 52 KB; the fully unrolled two-channel-block sliding-window conv
    bodies are the largest (~47 KB), the attention loops are 3-7 KB, the GEMM K loops ~8 KB; the opt-in GN forms of those conv bodies: <= 60 KB."""
    import shutil
    import tempfile

    import pytest

    from open_sora_amd.build import build_lib

    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("ROCm LLVM tools not found")
    lib_path = build_lib()
    loops = {}
    with tempfile.TemporaryDirectory() as d:
        lib = shutil.copy(lib_path, os.path.join(d, "libosk_hip.so"))
        subprocess.run([objdump, "--offloading", lib], capture_output=True, text=True, cwd=d)
        for co in sorted(os.path.join(d, f) for f in os.listdir(d) if "gfx950" in f):
            cur, mfma_at, back = None, [], []

            def close():
                # the K loop of a kernel = its SMALLEST backward-branch span that holds at least 32 MFMAs (the persistent kernels'
                # outer tile loop also spans the epilogue -- 100+ KB executed once per tile -- and is not what has to stay resident)
                spans = [e - b for b, e in back if sum(1 for a_ in mfma_at if b <= a_ < e) >= 32]
                if cur and spans:
                    loops[cur] = min(spans)

            for ln in subprocess.run([objdump, "-d", co], capture_output=True, text=True).stdout.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
                if m:
                    close()
                    cur, mfma_at, back = m.group(1), [], []
                    continue
                m = re.search(r"// ([0-9A-F]+):", ln)
                if not m or not cur:
                    continue
                addr = int(m.group(1), 16)
                if "v_mfma" in ln:
                    mfma_at.append(addr)
                m2 = re.match(r"\s+s_c?branch\w* (\d+)\s", ln)
                if m2 and int(m2.group(1)) >= 0x8000:                # 16-bit signed dword offset: backward
                    back.append((addr + 4 - (0x10000 - int(m2.group(1))) * 4, addr + 4))
            close()
    hand = {k: v for k, v in loops.items() if re.search(r"attn_asm|gemm256|convsw|conv256x", k)}
    assert len(hand) >= 12, sorted(loops)
    # the opt-in GN forms of the sliding-window conv (input GroupNorm + SiLU folded in: + 72 VALU per halo piece) are the two
    # largest loops of the library: still inside the cache, with less room
    gn = {k: hand.pop(k) for k in list(hand) if re.search(r"convsw2_kernelILb1E|convsw_kernelILi8ELb0ELb1E", k)}
    assert len(gn) == 2 and max(gn.values()) <= 60 * 1024, gn
    worst = max(hand.values())
    assert 30 * 1024 < worst <= 52 * 1024, {k[:60]: v for k, v in hand.items() if v == worst}
    for k, v in hand.items():
        if "attn_asm" in k:
            assert v <= 16 * 1024, (k, v)


def test_hbm_kernels_issue_their_loads_before_the_first_wait():
    """Round 5's rule for the HBM-bound kernels: all loads of a block are in flight before the first one is
    waited for.  hipcc sinks a load whose only use sits behind a predicate into that branch and waits there -- one memory round trip
    per piece (qknorm_rope_rows: 128 -> 84 us once the loads were unconditional with clamped indices).  Checked on the BUILT library:
    the longest run of global loads without a vmcnt wait between them, per kernel."""
    import shutil
    import tempfile

    import pytest

    from open_sora_amd.build import build_lib

    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("ROCm LLVM tools not found")
    want = {r"qknorm_rope_rows_kernelILi72ELi0ELi128E": 8,      # 9 span pieces + 5 cos / sin pieces (the compiler keeps 8 in flight)
            r"qknorm_rope_rows_kernelILi64ELi1ELi128E": 8,
            r"gemv_tasks_kernelILi4E": 8,                        # 8 weight rows per wave; 8 x elements per staging trip
            r"gemv_tasks_kernelILi8E": 4,
            r"ln_modulate_kernelILi3ELb0E": 3,                   # the row's three 1 KB pieces (XL: D = 1152)
            r"ln_modulate_kernelILi6ELb0E": 6}                   # 11B: D = 3072
    runs = {}
    with tempfile.TemporaryDirectory() as d:
        lib = shutil.copy(build_lib(), os.path.join(d, "libosk_hip.so"))
        subprocess.run([objdump, "--offloading", lib], capture_output=True, text=True, cwd=d)
        for co in sorted(os.path.join(d, f) for f in os.listdir(d) if "gfx950" in f):
            cur, run = None, 0
            for ln in subprocess.run([objdump, "-d", co], capture_output=True, text=True).stdout.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
                if m:
                    cur, run = next((k for k in want if re.search(k, m.group(1))), None), 0
                    continue
                if cur is None:
                    continue
                if re.match(r"\s+global_load_", ln):
                    run += 1
                    runs[cur] = max(runs.get(cur, 0), run)
                elif re.match(r"\s+s_waitcnt.*vmcnt", ln):
                    run = 0
    assert set(runs) == set(want), sorted(set(want) - set(runs))
    for k, n in want.items():
        assert runs[k] >= n, (k, runs[k], n)


def test_no_orphan_generated_files():
    made = set()
    for p in glob.glob(os.path.join(CSRC, "*.inc")):
        assert open(p).readline().startswith("// GENERATED by tools/"), p
        made.add(os.path.basename(p))
    used = set()
    for p in glob.glob(os.path.join(CSRC, "*.hip")):
        used |= set(re.findall(r'#include "([^"]+\.inc)"', open(p).read()))
    assert used <= made, f"missing generated files: {used - made}"
    assert made <= used, f"generated but never included: {made - used}"


def test_attention_schedule_invariants():
    """every LDS fragment read is waited for before its MFMA (the generator's in-order lgkmcnt bookkeeping), every
    body has the same MFMA count as the tile needs, and the asm never names a register outside its declared ranges."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_attn_asm as G

    for hd, nu, pv8, fast in ((72, 2, False, False), (72, 1, False, False), (128, 2, False, False), (72, 2, True, False),
                              (128, 2, True, False), (72, 2, False, True), (128, 2, False, True)):
        pv16 = hd == 72 and nu == 2 and not pv8      # the shipped head_dim-72 bodies: P.V on v_mfma_f32_16x16x32_bf16
        L = G.Layout(nu, hd, pv8, pv16=pv16)
        st = G.generate(L, fast=fast)
        text = "\n".join(st.lines)
        if fast:   # bounded reference: no max tracking, no rare path; lane-mask selects and ones-row writes only in the out-of-line
            # loader event code behind the loop (fast_events), one compare + branch per advance site in front of it
            loop, events = text.split("\n.L@@_ev", 1)
            assert "v_max" not in text and "s_cbranch_vccnz" not in text and "_rare" not in text
            assert "v_cndmask" not in loop and "ds_write_b32" not in loop
            assert loop.count("s_cbranch_scc1 .L@@_ev") == 7 and events.count("ds_write_b32") == 2 * 3   # 3 prologue + 2 x 2 body sites; V sites write 2 rows
        nm = len(re.findall(r"v_mfma_f32_32x32x16_bf16", text))
        nf = len(re.findall(r"v_mfma_f32_32x32x64_f8f6f4", text))
        nx = len(re.findall(r"v_mfma_f32_16x16x32_bf16", text))
        # prologue QK^T + two bodies (trailing P.V pairs + QK^T + the other P.V pairs each) + exit trailing pairs
        npk, npv = L.G.NPK, L.G.NPV
        if pv8:   # one fp8 MFMA per O^T row tile: P.V pairs = row tiles, one pair trails across the barrier
            assert (npk, npv) == ((10, 3) if hd == 72 else (18, 5))
            assert nm == npk * nu * 3 and nf == 2 * npv * nu + nu and nx == 0
        elif pv16:  # 5 row blocks of 16 x 2 key halves = 10 pairs of 2 * NU MFMAs, two pairs trail across the barrier
            assert (npk, npv) == (10, 10) and nf == 0
            assert nm == 3 * npk * nu and nx == 2 * npv * 2 * nu + 2 * 2 * nu
            assert text.count("v_permlane16_swap_b32") == 2 * (4 * 2 * nu) + (0 if fast else 2 * nu)   # 4 per (u, half) per body; rare paths: 1 per u
        else:
            assert (npk, npv) == ((10, 12) if hd == 72 else (18, 20)) and nf == 0 and nx == 0
            assert nm == npk * nu + 2 * ((npk + npv) * nu) + 2 * nu
        for m in re.finditer(r"\bv(\d+)\b", text):
            assert L.V_FIRST <= int(m.group(1)) < L.V_END, m.group(0)
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
            assert L.V_FIRST <= int(m.group(1)) and int(m.group(2)) < L.V_END, m.group(0)
        for m in re.finditer(r"\ba(\d+)\b", text):
            assert int(m.group(1)) < L.A_END
        for m in re.finditer(r"\ba\[(\d+):(\d+)\]", text):
            assert int(m.group(2)) < L.A_END
        for m in re.finditer(r"\bs(\d+)\b", text):
            assert G.S_FIRST <= int(m.group(1)) <= G.S_LAST, m.group(0)
        assert len(L.OPERANDS) <= 30  # inline-asm operand limit
        assert L.V_END <= 256 and L.V_END + L.A_END <= 512   # unified register file of one wave per SIMD
        for m in re.finditer(r"ds_read_b128 [^\n]* offset:(\d+)", text):
            assert int(m.group(1)) < 65536   # 16-bit LDS immediate


def test_asm_accumulators_are_values_the_compiler_knows():
    """The hand-scheduled loops leave their accumulators in fixed AGPRs.  Until round 4 the C++ epilogues read them back with
    asm statements that NAMED the registers while the loop statement merely clobbered a0..a255 -- nothing told the compiler that
    they were live, and under pressure the allocator parked its own values there (a wider conv epilogue made it use a0..a4 in
    `convsw_kernel<8, UP>`: first row pair of every wave tile wrong); round 4 guarded that with a scan of the compiler's
    assembly for AGPR writes.  Round 5 removed the cause (csrc/acc_quads.h, tools/gen_acc_quads.py): an empty asm statement
    behind every loop statement lists the accumulator quads as OUTPUTS in their physical registers, the attention kernels' Q
    fragments are ordinary values bound as loop INPUTS in theirs, and epilogues read `"a"(quad[i])` operands.  Checked here on
    the sources: no wrapper names an AGPR any more, every loop statement is followed by the binding, the generators emit the
    register map the wrappers static_assert, and the binding header is what its generator writes."""
    files = ["gemm256x.hip", "gemm256p.hip", "gemm256.hip", "conv3d_256.hip", "attention_asm72.hip", "attention_asm72w.hip",
             "attention_asm72p8.hip", "attention_asm128.hip", "attention_asm128p8.hip"]
    for name in files + ["gemm_epilogue.h", "gemm_epilogue16.h"]:
        src = open(os.path.join(CSRC, name)).read()
        code = "\n".join(ln.split("//")[0] for ln in src.split("\n"))
        assert not re.search(r"v_accvgpr_\w+ [^\n\"]*\ba\d+\b", code), f"{name}: an asm statement names an AGPR"
        assert not re.search(r"\b\w+_(AR|OR|QW)\d", code), f"{name}: a register-named read / write macro is back"
    for name in files:
        src = open(os.path.join(CSRC, name)).read()
        n_loops = len(re.findall(r'#include "\w+(?:_body\w*|_n2_[vf]0|72w_f0)\.inc"', src))
        assert n_loops >= 1, name
        # one binding per kernel body (a constexpr chain of alternative loop statements shares the one behind it)
        assert len(re.findall(r"OSK_AQ_OUT_0_\d+\(|OSKCX_ACC\(NBJ", src)) >= 1, name
        assert "static_assert(OSK" in src, f"{name}: the generated register map is not asserted"
        if name.startswith("attention_asm"):
            assert src.count("OSK_AQ_IN_") >= 2, f"{name}: the Q fragments are not bound as loop inputs"
    regs = open(os.path.join(CSRC, "attention_asm_regs.inc")).read() + open(os.path.join(CSRC, "gemm256x_regs.inc")).read()
    assert "accvgpr" not in regs and "OSKX_ACC_QUADS 64" in regs and "OSK72W_AQ0 160" in regs
    import subprocess as sp
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ)
        gen = open(os.path.join(TOOLS, "gen_acc_quads.py")).read().replace('os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "open_sora_amd", "csrc", "acc_quads.h")',
                                                                           repr(os.path.join(d, "acc_quads.h")))
        sp.run([sys.executable, "-c", gen], check=True, env=env, capture_output=True)
        assert open(os.path.join(d, "acc_quads.h")).read() == open(os.path.join(CSRC, "acc_quads.h")).read(), "acc_quads.h is stale: re-run tools/gen_acc_quads.py"


def test_no_compiler_instruction_writes_an_agpr_in_the_gemm_kernels(tmp_path):
    """the accumulators become compiler-visible through an empty asm statement BEHIND the K-loop statement, which only
    clobbers a0..a255 -- nothing in the constraints forbids the compiler to park a value of its own in an AGPR between the two
    statements.  The by-construction argument is that it has no reason to (its values live in VGPRs, spills go to scratch); this test
    pins it on the compiler's actual output: in the device assembly of gemm256x.hip (the hottest wrapper, 256 accumulators + ~250
    VGPRs: the highest register pressure of the library) no instruction OUTSIDE the inline-asm blocks has an AGPR destination."""
    import shutil
    import subprocess as sp

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    from open_sora_amd.build import flags_for

    src = os.path.join(CSRC, "gemm256x.hip")
    out = str(tmp_path / "gemm256x.s")
    flags = [f for f in flags_for(src) if f != "-fPIC"]
    r = sp.run([hipcc, *flags, "--cuda-device-only", "-S", src, "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    # hipcc brackets every asm statement with ;;#ASMSTART / ;;#ASMEND.  The K-loop statement is the one that contains MFMAs; the empty
    # statement that follows it is the binding.  What the compiler schedules BETWEEN the two (scalar compares, SGPR-spill reloads ...)
    # must not touch an AGPR: until the binding the accumulators are, to the compiler, dead registers it could use as scratch.
    lines = open(out).read().split("\n")
    n_loops, between, i = 0, [], 0
    while i < len(lines):
        if lines[i].strip().startswith(";;#ASMSTART"):
            j = i + 1
            has_mfma = False
            while not lines[j].strip().startswith(";;#ASMEND"):
                has_mfma = has_mfma or lines[j].lstrip().startswith("v_mfma")
                j += 1
            if has_mfma:                                   # a K-loop statement: collect up to the next asm statement (the binding)
                n_loops += 1
                k = j + 1
                while not lines[k].strip().startswith(";;#ASMSTART"):
                    t = lines[k].split(";")[0].strip()
                    if t and not t.endswith(":") and not t.startswith("."):
                        between.append(t)
                    k += 1
                assert lines[k + 1].strip().startswith(";;#ASMEND"), "the statement behind a K loop is not the empty accumulator binding"
            i = j
        i += 1
    assert n_loops >= 4, "the assembly does not look like the GEMM wrappers'"
    bad = [t for t in between if re.search(r"\ba\d+\b|\ba\[\d", t) or "accvgpr" in t]
    assert not bad, f"the compiler touches AGPRs between a K loop and its accumulator binding: {bad[:5]}"
