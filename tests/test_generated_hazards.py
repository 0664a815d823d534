"""Static hazard check ("race screen") of the generated GEMM / conv K loops (open_sora_amd/csrc/*_body*.inc, *_segment*.inc).

A dataflow pass over each asm body's control-flow graph (union at joins, iterated to a fixed point) tracks
  * registers with an LDS read in flight (ds_read_* destinations until `s_waitcnt lgkmcnt(0)`),
  * LDS stages with an LDS-DMA fill in flight ((operand, stage) from the M0 value of each `global_load_lds_dwordx4`, until
    `s_waitcnt vmcnt(0)` FOLLOWED BY `s_barrier`: the fill of another wave is only known to have landed behind the barrier),
  * LDS stages read since the last barrier,
and reports
  RAW-reg : an MFMA / VALU instruction reads a register whose ds_read has not been waited for,
  RAW-lds : a fragment read from a stage whose fill is still in flight,
  WAR-lds : an LDS-DMA fill issued into a stage that was read since the last barrier (a slower wave may still be reading it),
  M0      : an LDS-DMA instruction without a fresh M0 write, or directly behind it (the hardware needs one instruction between).
LDS operations retire in order, so `s_waitcnt lgkmcnt(N)` retires all but the N most recent reads (the attention bodies count
their waits).  The flash-attention bodies (tools/gen_attn_asm.py) get the register and M0 checks only: their K / V^T rings are
not two symmetric stages.
The schedules are hand-designed in tools/gen_*_asm.py and validated on the GPU; this guards future edits of the generators
on a machine without one."""
import glob
import os
import re

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "open_sora_amd", "csrc")
BODIES = sorted(p for p in glob.glob(os.path.join(CSRC, "*.inc"))
                if ("_body" in p or "_segment" in p) and "attention" not in os.path.basename(p)
                and "convsw" not in os.path.basename(p))   # the sliding-window conv has its own executable model: tests/test_conv_sw_model.py
ATTN_BODIES = sorted(p for p in glob.glob(os.path.join(CSRC, "attention_asm*_n*_[vf]0.inc")) +   # general (v0) and fast (f0) bodies
                     glob.glob(os.path.join(CSRC, "attention_asm72w_f0.inc")))                   # + the wide layout's
S_ADST, S_WDST = "s45", "s46"
A_STAGE = 32768


def parse(path):
    ins = []
    for raw in open(path):
        raw = raw.strip()
        if not raw.startswith('"'):
            continue
        for text in raw.strip('"').replace("\\n", "\n").split("\n"):
            text = text.strip()
            if text:
                ins.append(text)
    return ins


def regs_of(tok):
    """'v[176:179]' -> ['v176', ..], 'v12' -> ['v12'], 'a[0:3]' -> a-registers, '%5' -> ['%5'], anything else -> []"""
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return ["%s%d" % (m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    if re.fullmatch(r"[va]\d+", tok) or re.fullmatch(r"%\d+", tok):
        return [tok]
    return []


def lint(path, stages=True):
    ins = parse(path)
    labels = {t[:-1]: i for i, t in enumerate(ins) if t.endswith(":")}
    w4_style = any(t.startswith("v_xor_b32_e32") and t.rstrip().endswith("%0") for t in ins)
    cls = {"%0": "A", "%1": "W"} if w4_style else {**{"%%%d" % i: "A" for i in range(4)}, **{"%%%d" % i: "W" for i in range(4, 8)}}
    for t in ins:                                    # address registers derived by XOR inherit the operand class
        m = re.match(r"v_xor_b32_e32 (v\d+), \d+, (%\d+)", t)
        if m and m.group(2) in cls:
            cls[m.group(1)] = cls[m.group(2)]
    w_imm = [int(m.group(1)) for t in ins for m in [re.match(r"s_add_u32 m0, %s, (\d+)" % S_WDST, t)] if m]
    w_stage = 16384 if w_imm and max(w_imm) < 32768 else 32768
    stage_of = lambda op, off: off // (A_STAGE if op == "A" else w_stage)

    def succ(i):
        t = ins[i]
        out = []
        m = re.match(r"s_(c?branch\w*) (\S+)", t)
        if m:
            tgt = m.group(2)
            assert tgt in labels, (path, tgt)
            out.append(labels[tgt])
            if m.group(1) == "branch":
                return out
        if i + 1 < len(ins):
            out.append(i + 1)
        return out

    # state: (LDS reads in flight: tuple of destination sets in issue order, pending_dma stages, dma_landed_locally,
    #         read_since_barrier, m0, m0_age)
    empty = ((), frozenset(), frozenset(), frozenset(), None, 9)
    state_in = {0: empty}
    work = [0]
    errors = set()

    def step(i, st):
        pq, dma, landed, rsb, m0, age = st
        pend = set().union(*pq) if pq else set()
        t = ins[i]
        if t.endswith(":"):
            return st
        op = t.split()[0]
        args = [a.strip() for a in t[len(op):].split(",")]
        if op.startswith("ds_read"):
            dst = regs_of(args[0])
            addr = args[1].split()[0]
            off = int(re.search(r"offset:(\d+)", t).group(1)) if "offset:" in t else 0
            if op == "ds_read_b128" and stages:
                assert addr in cls, (path, t)
                key = (cls[addr], stage_of(cls[addr], off))
                if key in dma or key in landed:
                    errors.add(("RAW-lds", i, t))
                rsb = rsb | {key}
            for r in regs_of(addr):
                if r in pend:
                    errors.add(("RAW-reg", i, t))
            pq = pq + (frozenset(dst),)
        elif op in ("s_add_u32", "s_mov_b32") and args[0] == "m0":
            base, imm = args[1], int(args[2]) if len(args) > 2 and args[2].isdigit() else 0
            m0 = ("A" if base == S_ADST else "W", stage_of("A" if base == S_ADST else "W", imm)) if stages else ("any", 0)
            age = 0
            return (pq, dma, landed, rsb, m0, age)
        elif op == "global_load_lds_dwordx4":
            if m0 is None or age < 1:
                errors.add(("M0", i, t))
            elif stages:
                if m0 in rsb:
                    errors.add(("WAR-lds", i, t))
                dma = dma | {m0}
            for r in regs_of(args[0]):
                if r in pend:
                    errors.add(("RAW-reg", i, t))
            m0 = None
        elif op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                pq = pq[len(pq) - n:] if n else ()
            if "vmcnt(0)" in t:
                landed, dma = landed | dma, frozenset()
        elif op == "s_barrier":
            landed, rsb = frozenset(), frozenset()
        elif op.startswith("v_mfma"):
            for a in args[1:3]:
                for r in regs_of(a):
                    if r in pend:
                        errors.add(("RAW-reg", i, t))
        elif op.startswith("v_"):
            srcs = args if op.startswith("v_permlane") else args[1:]      # permlane swaps read both operands
            for a in srcs:
                for r in regs_of(a.split()[0] if a else a):
                    if r in pend:
                        errors.add(("RAW-reg", i, t))
        return (pq, frozenset(dma), frozenset(landed), frozenset(rsb), m0, min(age + 1, 9))

    def join(a, b):
        if a is None:
            return b
        qa, qb = a[0], b[0]                       # align the in-flight reads from the most recent one backwards
        n = max(len(qa), len(qb))
        qa, qb = (frozenset(),) * (n - len(qa)) + qa, (frozenset(),) * (n - len(qb)) + qb
        pq = tuple(x | y for x, y in zip(qa, qb))
        return (pq, a[1] | b[1], a[2] | b[2], a[3] | b[3], a[4] if a[4] == b[4] else None, min(a[5], b[5]))

    while work:
        i = work.pop()
        out = step(i, state_in[i])
        for j in succ(i):
            merged = join(state_in.get(j), out)
            if merged != state_in.get(j):
                state_in[j] = merged
                work.append(j)
    return sorted(errors, key=lambda e: e[1]), len(ins)


@pytest.mark.parametrize("path", BODIES, ids=[os.path.basename(p) for p in BODIES])
def test_generated_k_loop_has_no_static_hazard(path):
    errors, n = lint(path)
    assert n > 100
    assert not errors, "%s: %d hazards, first: %s" % (os.path.basename(path), len(errors), errors[:5])


@pytest.mark.parametrize("path", ATTN_BODIES, ids=[os.path.basename(p) for p in ATTN_BODIES])
def test_generated_attention_loop_has_no_register_hazard(path):
    errors, n = lint(path, stages=False)
    assert n > 500
    assert not errors, "%s: %d hazards, first: %s" % (os.path.basename(path), len(errors), errors[:5])


def test_the_linter_sees_a_missing_wait(tmp_path):
    """teeth: drop every fragment-read wait from a shipped body -> RAW-reg reports; drop the barriers -> WAR-lds reports"""
    src = os.path.join(CSRC, "gemm256x_body.inc")
    text = open(src).read()
    p1 = tmp_path / "nowait_body.inc"
    p1.write_text(text.replace("lgkmcnt(0)", "lgkmcnt(15)"))
    assert any(e[0] == "RAW-reg" for e in lint(str(p1))[0])
    p2 = tmp_path / "nobarrier_body.inc"
    p2.write_text(text.replace('"  s_barrier\\n"\n', ""))
    kinds = {e[0] for e in lint(str(p2))[0]}
    assert "WAR-lds" in kinds or "RAW-lds" in kinds
