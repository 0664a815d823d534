"""CPU checks of the sliding-window CausalConv3d K loop (tools/gen_conv_sw_asm.py -> csrc/convsw_body_n*.inc): the generated
instruction stream is executed symbolically (hazards, pointer arithmetic, completeness of the accumulation) and numerically,
lane by lane, against a float64 convolution (tests/conv_sw_emulator.py).  No GPU: this is what guards edits of the generator
and of the wrapper's lane formulas on a machine without one; tests/test_gpu_vae.py compares the kernel itself with the oracle."""
import numpy as np
import pytest

from tests import conv_sw_emulator as E
from tests.test_lds_fragment_maps import conflicts


@pytest.mark.parametrize("up", [False, True])
@pytest.mark.parametrize("cin", [64, 128, 256])
@pytest.mark.parametrize("nbj", [8, 4])
def test_schedule_is_hazard_free_and_complete(nbj, cin, up):
    for wave in (0, 3):
        r = E.check_schedule(nbj, cin, wave, up)
        assert r["barriers"] == 1 + (54 // E.G.Cfg(nbj).BAR) * (cin // 64)


@pytest.mark.parametrize("cin", [64, 128, 256])
def test_two_frame_schedule_is_hazard_free_and_complete(cin):
    for wave in (0, 3):
        r = E.check_schedule(8, cin, wave, f2=True)
        assert r["barriers"] == 1 + 54 * (cin // 64)


@pytest.mark.parametrize("up", [False, True])
@pytest.mark.parametrize("nbj", [8, 4])
def test_pieces_of_the_four_waves_cover_a_stage_and_a_frame_slot(nbj, up):
    assert E.piece_coverage(nbj, up) == (True, True)
    assert E.piece_coverage(8, f2=True) == (True, True)


def test_lds_budget():
    c = E.G.Cfg(8)
    assert c.SMEM <= 160 * 1024 and c.HALO == 3 * 21 * 1024 and E.G.Cfg(8, True).HALO == 3 * 7 * 1024
    assert max(c.a_offset(tap, i)[0] for tap in range(27) for i in range(8)) + 16 * 64 + 64 <= 65536   # 16-bit ds_read immediates


def test_fragment_reads_are_bank_conflict_free():
    """halo: 64-byte voxels, chunk c of halo column ww at position c ^ ((ww >> 1) & 3); lane l reads brick column l % 16 shifted by
    the tap's dw, chunk l / 16 -- every tap, every row block, both voxel halves; weights: 64-byte rows, key (row >> 1) & 3"""
    c, cu = E.G.Cfg(8), E.G.Cfg(8, True)
    for tap in range(27):
        for i in range(8):
            off, dw = c.a_offset(tap, i)
            offu, _ = cu.a_offset(tap, i)
            for wm in range(2):
                assert conflicts(lambda l: (144 * wm + (l & 15)) * 64 + (((l >> 4) ^ ((((l & 15) + dw) >> 1) & 3)) << 4) + off) == 0
                # upsampled form: lane pairs share a source voxel (same address: a broadcast, no conflict)
                ww = lambda l: (((l & 15) + dw - 1) >> 1) + 1
                assert conflicts(lambda l: (40 * wm + ww(l)) * 64 + (((l >> 4) ^ int(E.up_key(ww(l)))) << 4) + offu) == 0
    for base in range(0, 256, 16):
        assert conflicts(lambda l: (base + (l & 15)) * 64 + (((l >> 4) ^ (((l & 15) >> 1) & 3)) << 4)) == 0
    assert conflicts(lambda l: (l & 15) * 64 + ((l >> 4) << 4)) > 0          # the check has teeth: no swizzle -> conflicts


def _case(seed, T, H, W, cin, cout):
    rng = np.random.default_rng(seed)
    wrs = (27 * cin + 63) // 64 * 64
    x = rng.standard_normal((T, H, W, cin)).astype(np.float32)
    w = np.zeros((cout, wrs), np.float32)
    w[:, : 27 * cin] = rng.standard_normal((cout, 27 * cin)).astype(np.float32) * (27 * cin) ** -0.5
    xb = (x.view(np.uint32) >> 16).astype(np.uint16)             # truncate to bf16: exact inputs for both sides
    wb = (w.view(np.uint32) >> 16).astype(np.uint16)
    x = (xb.astype(np.uint32) << 16).view(np.float32)
    w = (wb.astype(np.uint32) << 16).view(np.float32)
    return x, w, xb, wb, (T, H, W, cin, cout, wrs)


@pytest.mark.parametrize("nbj,cin,cout,tile,up", [
    (4, 64, 128, (0, 0, 0, 0), (False, False)),     # first frame (causal clamp), top-left corner (replicate clamp), one body iteration
    (4, 128, 128, (2, 1, 1, 0), (False, False)),    # interior brick, two body iterations (the loop's back edge, channel-block wrap)
    (8, 64, 320, (1, 2, 1, 256), (False, False)),   # bottom edge, 256-wide tile whose last rows lie beyond Cout (clamped weight rows)
    (8, 128, 256, (2, 1, 2, 0), (False, False)),    # right edge, two iterations, 256-wide
    (8, 64, 256, (0, 0, 0, 0), (True, True)),       # upsample T, H, W: output frame 0 (all taps read source frame 0), corner
    (8, 128, 256, (3, 2, 3, 0), (True, True)),      # output frame 3 = source frames 1, 1, 2; interior brick of the 96 x 96 output
    (4, 64, 128, (4, 5, 5, 0), (True, True)),       # last brick row / column (replicate clamp at the far sides), output frame 4
    (8, 64, 256, (1, 3, 0, 0), (False, True)),      # H, W upsample only: frames map one to one
])
def test_generated_stream_computes_the_convolution(nbj, cin, cout, tile, up):
    x, w, xb, wb, geom = _case(3, 3, 48, 48, cin, cout)
    got = E.emulate_tile(nbj, xb, wb, geom, tile, up)
    ncols = min(32 * nbj, cout - tile[3])
    ref = E.reference_tile(x, w, geom, tile, ncols, up)
    err = np.abs(got[:, :ncols] - ref).max()
    assert err <= 2e-4 * max(1.0, np.abs(ref).max()), err


@pytest.mark.parametrize("cin,tile", [
    (64, (0, 0, 0, 0)),       # frames 0, 1: slots 0, 1 hold frame 0 twice (causal clamp); corner brick
    (128, (1, 1, 2, 0)),      # frames 2, 3, two body iterations, right-edge brick
    (64, (2, 2, 1, 0)),       # last pair of an odd T = 5: frame 4 + a frame that does not exist (its rows are never stored)
])
def test_two_frame_stream_computes_the_convolution(cin, tile):
    x, w, xb, wb, geom = _case(5, 5, 48, 48, cin, 128)
    got = E.emulate_tile(8, xb, wb, geom, tile, f2=True)
    for f in range(2):
        t = 2 * tile[0] + f
        if t >= 5:
            continue
        ref = E.reference_tile(x, w, geom, (t, tile[1], tile[2], 0), 128)
        err = np.abs(got[256 * f: 256 * f + 256] - ref).max()
        assert err <= 2e-4 * max(1.0, np.abs(ref).max()), (f, err)


# ---- GN form: the input GroupNorm + SiLU folded into the halo refill (register-staged pieces, transform in the MFMA gaps) ----------

@pytest.mark.parametrize("f2", [False, True])
@pytest.mark.parametrize("cin", [64, 128, 256])
def test_gn_form_schedule_is_hazard_free_and_complete(cin, f2):
    """+ the GN-specific checks of conv_sw_emulator.check_schedule: a piece is transformed only after its load has been waited for,
    with the (scale, shift) rows of ITS channel block resident (no table load under a half-transformed piece), a register quad is
    not re-loaded before its piece is written, an LDS slot is written only behind the barrier after its last reader and read only
    behind a barrier after the write retired, every slot gets all six pieces of a wave"""
    for wave in (0, 3):
        r = E.check_schedule(8, cin, wave, f2=f2, gn=True)
        assert r["barriers"] == 1 + 54 * (cin // 64)


def test_gn_form_checker_has_teeth():
    """moving a transform window behind its slot's first reader, or onto another block's table rows, is caught"""
    G = E.G
    for f2, windows, what in [
            (True, {0: (2, 2.0, 9.5)}, "halo slot read"),                 # slot 2's pieces still being written when tap 9 reads it
            (False, {9: (0, 10.0, 19.5)}, "rows of another channel"),     # block B's pieces under block A's rows (table switch at 10.55)
    ]:
        G.GN_EXP["windows"] = {f2: windows}
        try:
            with pytest.raises(AssertionError) as ei:
                E.check_schedule(8, 128, 0, f2=f2, gn=True)
            assert what in str(ei.value) or "overlap" in str(ei.value), (what, str(ei.value)[:200])
        finally:
            G.GN_EXP.clear()


def test_gn_form_transcendental_results_are_not_read_by_the_next_instruction():
    """gfx940 / gfx950: a non-transcendental VALU instruction must not read a v_exp / v_rcp result in the very next issue slot
    (the assembler does not pad inline asm); and the transform never touches a fragment or accumulator register"""
    import re
    for f2 in (False, True):
        c = E.G.Cfg(8, f2=f2, gn=True)
        ops = [o for o in E.G.generate(c) if o.kind != "L"]
        for a, b in zip(ops, ops[1:]):
            m = re.match(r"v_(exp|rcp)_f32_e32 v(\d+),", a.text)
            if m and b.kind == "V":
                assert not re.search(r"\bv%s\b" % m.group(2), b.text.split(",", 1)[1]), (a.text, b.text)
        body_valu = [o for o in ops if o.kind == "V"]
        lo = c.VST
        for o in body_valu:
            if o.meta["quad"].__class__ is tuple:      # the prologue's pieces sit in the (still unused) fragment registers
                continue
            regs = [int(r) for r in re.findall(r"\bv(\d+)\b", o.text)]
            assert all(lo <= r < c.V0 + c.VN for r in regs), o.text


@pytest.mark.parametrize("f2,cin,cout,tile,T", [
    (False, 64, 320, (1, 2, 1, 256), 3),     # 256-channel tiles: ragged last channel tile, bottom edge
    (False, 128, 256, (2, 1, 2, 0), 3),      # two body iterations (the table pointer walks four channel blocks), right edge
    (True, 64, 128, (0, 0, 0, 0), 5),        # two-frame form: frames 0, 1 (causal clamp), corner brick
    (True, 128, 128, (2, 2, 1, 0), 5),       # last pair of an odd T: the second frame does not exist
])
def test_gn_form_stream_computes_the_convolution_of_the_normalised_input(f2, cin, cout, tile, T):
    x, w, xb, wb, geom = _case(7, T, 48, 48, cin, cout)
    rng = np.random.default_rng(cin + cout)
    scale = (0.5 + rng.random(cin)).astype(np.float32)
    shift = (rng.standard_normal(cin) * 0.5).astype(np.float32)
    got = E.emulate_tile(8, xb, wb, geom, tile, f2=f2, gn_tab=E.gn_table(scale, shift))
    xt = E.gn_silu(x, scale, shift)          # bit-equal conv inputs on both sides: the same numpy operations as the emulated VALU
    if f2:
        for f in range(2):
            t = 2 * tile[0] + f
            if t < T:
                ref = E.reference_tile(xt, w, geom, (t, tile[1], tile[2], 0), 128)
                err = np.abs(got[256 * f: 256 * f + 256] - ref).max()
                assert err <= 2e-4 * max(1.0, np.abs(ref).max()), (f, err)
    else:
        ncols = min(256, cout - tile[3])
        ref = E.reference_tile(xt, w, geom, tile, ncols)
        err = np.abs(got[:, :ncols] - ref).max()
        assert err <= 2e-4 * max(1.0, np.abs(ref).max()), err


def test_gn_silu_model_matches_torch():
    """conv_sw_emulator.gn_silu (what the emulated instructions compute) against torch's bf16 GroupNorm-apply + SiLU semantics"""
    import torch
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((4096, 32)) * 2).astype(np.float32)
    x = ((x.view(np.uint32) >> 16) << 16).view(np.float32)
    a, d = (0.5 + rng.random(32)).astype(np.float32), rng.standard_normal(32).astype(np.float32)
    got = E.gn_silu(x, a, d)
    y = (torch.from_numpy(x) * torch.from_numpy(a) + torch.from_numpy(d)).to(torch.bfloat16).float()
    want = torch.nn.functional.silu(y).to(torch.bfloat16).float().numpy()
    assert (got != want).mean() < 2e-2 and np.abs(got - want).max() <= 2.0 ** -7 * np.abs(want).max()


def test_generated_bodies_name_only_declared_registers():
    """every v / a / s register a body names is inside the clobber list its asm statement declares (csrc/convsw_regs.inc) -- a
    register outside it would silently corrupt the compiler's state around the asm statement"""
    import os
    import re

    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "open_sora_amd", "csrc")
    regs = open(os.path.join(csrc, "convsw_regs.inc")).read()
    clob = {n: set(re.findall(r'"([vas]\d+)"', re.search(r"#define OSKSW%s_CLOBBERS (.*)" % n, regs).group(1))) for n in (256, 128, "G256", "GF128")}
    used_by = {"convsw_body_n256.inc": 256, "convswu_body_n256.inc": 256, "convswf_body_n128.inc": 256,   # the two-frame form: 128 x 128 wave tiles
               "convsw_body_n128.inc": 128, "convswu_body_n128.inc": 128, "convswg_body_n256.inc": "G256", "convswgf_body_n128.inc": "GF128"}
    for name, n in used_by.items():
        text = open(os.path.join(csrc, name)).read()
        named = set()
        for kind, lo, hi in re.findall(r"\b([vas])\[(\d+):(\d+)\]", text):
            named |= {"%s%d" % (kind, i) for i in range(int(lo), int(hi) + 1)}
        named |= {"%s%s" % (k, i) for k, i in re.findall(r"\b([vas])(\d+)\b", text)}
        assert named and named <= clob[n], (name, sorted(named - clob[n])[:8])
        assert len(re.findall(r"v_mfma_f32_16x16x32_bf16", text)) == 54 * 8 * ((n if isinstance(n, int) else 256) // 32)
