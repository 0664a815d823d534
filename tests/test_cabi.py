"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/osk.h declares
(no compute calls — there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "osk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(osk_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_all_declared_symbols():
    from open_sora_amd.build import build_lib

    path = build_lib()
    assert os.path.isfile(path)
    lib = ctypes.CDLL(path)
    names = _declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/osk.h but not exported"
    lib.osk_arch.restype = ctypes.c_char_p
    assert lib.osk_arch() == b"gfx950"
    assert lib.osk_abi_version() == 2


def test_binding_signatures_cover_header():
    from open_sora_amd import _C

    assert set(_declared_symbols()) == set(_C.SIGNATURES)


def test_invalid_arguments_return_status_not_crash():
    """Argument validation happens before any HIP call: status < 0, nothing launched."""
    from open_sora_amd import _C

    lib = _C.lib
    # K not a multiple of 64
    st = lib.osk_gemm_bf16(16, 0, 8, 1, 16, 8, None, 16, 0, 8, 1, None, None, 0, 1, 8, 8, 8, 0, None)
    assert st < 0
    # unsupported head_dim
    st = lib.osk_attention_fwd_bf16(16, 0, 8, 16, 0, 0, 8, 16, 0, 16, 0, 8, None, 1, 1, 8, 1, 8, 48, 1.0, 0, 0, None)
    assert st < 0
    st = lib.osk_ln_modulate_bf16(None, 0, 0, None, 0, 0, None, None, 0, 1, 1, 8, 1e-6, None)
    assert st < 0


def test_product_path_does_not_import_oracle():
    """open_sora_amd/ must never import oracle/ (the oracle is test infrastructure)."""
    pkg = os.path.join(ROOT, "open_sora_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{f} imports oracle"


def test_osk_trace_records_calls(monkeypatch):
    """OSK_TRACE=1: the binding records every C-ABI call (entry point, scalars, buffers by order of first appearance and their
    alignment) before forwarding it -- the debugging aid behind tools/diff_traces.py.  No GPU needed: an argument check that
    fails still goes through the tracer."""
    import importlib

    monkeypatch.setenv("OSK_TRACE", "1")
    import open_sora_amd._C as C0
    C1 = importlib.reload(C0)
    try:
        assert C1.lib.osk_abi_version() == 2            # not in SIGNATURES' argument-carrying set: passes through
        n0 = len(C1.TRACE_LOG)
        rc = C1.lib.osk_blend_bf16(None, None, 1, 1, 1, 0, 1, None)      # NULL pointers -> invalid argument, nothing launched
        assert rc != 0
        rec = C1.TRACE_LOG[n0:]
        assert rec and rec[-1][0] == "osk_blend_bf16" and rec[-1][1] == "NULL" and rec[-1][3:8] == (1, 1, 1, 0, 1)
        rc = C1.lib.osk_blend_bf16(0x1000, 0x2010, 1, 1, 1, -1, 1, None)  # extent < 0 -> invalid argument
        assert rc != 0 and C1.TRACE_LOG[-1][1:3] == ("p0@00", "p1@10")
    finally:
        monkeypatch.delenv("OSK_TRACE")
        importlib.reload(C1)


def test_attention_launch_shape_is_chosen_by_rounds_of_the_chip():
    """osk_attention_launch_shape reports what osk_attention_fwd_bounded_bf16 launches, from the same selection code (host-only: no
    GPU needed; 256 CUs assumed without a device).  The wide 512-row layout only where its work units fill the chip at least as
    well as 256-row ones (Lq = 1024 with B x H = 16 and no workspace must keep 64 workgroups, not 32)."""
    from open_sora_amd import _C as C1

    ws = C1.lib.osk_attention_workspace_bytes()
    assert C1.attention_launch_shape(3, 16, 16896, 1, 16896, 72, 12.9, ws)[1] == 512        # the timed denoise step
    assert C1.attention_launch_shape(3, 16, 16896, 1, 16896, 72, 0.0, ws)[1] == 256         # no bound: the 256-row general body
    assert C1.attention_launch_shape(3, 16, 16896, 1, 16896, 128, 17.0, ws)[1] == 256       # head_dim 128 has no wide layout
    assert C1.attention_launch_shape(1, 16, 1024, 1, 1024, 72, 12.9, 0) == (1, 256)          # few units, nothing to split them with
    parts, rows = C1.attention_launch_shape(1, 16, 16896, 1, 16896, 72, 12.9, ws)            # B = 1: 528 wide units = 2 rounds + 16
    assert rows == 512 and parts == 8
    assert C1.lib.osk_attention_tail_split_factor(1, 16, 16896, 1, 16896, 72, ws) == C1.attention_launch_shape(1, 16, 16896, 1, 16896, 72, 0.0, ws)[0]


def test_gemm_tile_choice_follows_the_measured_rates():
    """osk_gemm_tile_choice reports the tile kernel osk_gemm_bf16 launches (host-only; 256 CUs assumed without a device).  The
    estimate was calibrated in round 5 against a same-process A/B of the three kernels (profiles/r05i_gemm_tile_ab.jsonl): the cases
    below are the measured winners at the XL shapes -- CFG batch 3 and 1, and the rows sequence-parallel ranks hold."""
    from open_sora_amd import _C as C1

    L = 16896
    pick = C1.lib.osk_gemm_tile_choice
    for n, k in ((1152, 1152), (1152, 4608), (1152, 5760), (3456, 1152), (4608, 1152), (8064, 1152)):
        assert pick(3 * L, n, k) == 2 and pick(L, n, k) == 2, (n, k)          # B = 3 and B = 1: 256 x 256 everywhere (B = 1, N = 1152:
        assert pick(3 * L // 2, n, k) == 2 and pick(3 * L // 4, n, k) == 2     # two rounds of 256 x 256 beat three of 256 x 128)
    for n, k in ((1152, 1152), (1152, 4608), (1152, 5760)):
        assert pick(3 * L // 8, n, k) == 1 and pick(L // 4, n, k) == 1, (n, k)  # 25 / 17 row tiles x 5: half the chip idle with 256 x 256
    assert pick(3 * L // 8, 8064, 1152) == 2
    assert C1.lib.osk_gemm_tile_override(3) != 0 and C1.lib.osk_gemm_tile_override(-1) == 0   # invalid kind refused; estimate restored
    assert C1.lib.osk_attention_rows_override(128) != 0 and C1.lib.osk_attention_rows_override(0) == 0
