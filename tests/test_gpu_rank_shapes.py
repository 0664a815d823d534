"""Parity of the attention bodies a sequence-parallel RANK actually runs at BASELINE's long-sequence configurations
(VERDICT r5 missing #2 / weak P-i): `seqpar.SeqPar.attention` hands the block's score bound down, so a rank's launch is the
bounded FAST body (wide 512-row units for head_dim 72 where they fill the chip) -- or, in fp8 mode, the fp8 P.V body -- over
P key segments of L/P keys, NOT the general body `test_attention_sequence_parallel_rank_shape_720p` exercises.

Shapes (the reference's SP entry /root/reference/opensora/models/mmdit/distributed.py:413-422,580-683):
    configs[3]  XL,  51 x 720p latent: L = 184,112 = 8 x 23,014   (head_dim 72, 16 heads)
    configs[4]  11B, 64 x 720p latent: L = 230,912 = 8 x 28,864   (head_dim 128, 24 heads; bf16 and fp8 P.V)
    768 px      11B, the reference's shipped SP = 8 workload: L = 76,544 = 8 x 9,568, CFG batch 3
both exchange layouts: "allgather" (my L/P query rows, all heads) and "heads" (head exchange: P x B query batches of L/P
rows that share B key sets -- kv_batches -- with H/P heads).  Full size on the key axis (where 32-bit loader offsets and the
segment-event arithmetic would break), a few thousand query rows per batch item so that whole 512-row units AND a ragged one exist.
Checked against fp64 on sample rows (every unit kind) and through the constant-V property on all rows."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
BF = torch.bfloat16
F8 = torch.float8_e4m3fn
LOG2E = 1.4426950408889634


def _qk(P, Bq, Bkv, Lq, Lloc, H, hd, seed):
    """q, k shaped like the model's: unit-RMS rows per head (QK-norm), q carrying hd^-1/2 log2 e (q_prescaled)"""
    g = torch.Generator(device=DEV).manual_seed(seed)
    D = H * hd

    def unit_rms(t):
        t = t.view(*t.shape[:-1], H, hd)
        return (t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-6)).view(*t.shape[:-2], D)

    q = (unit_rms(torch.randn(Bq, Lq, D, device=DEV, generator=g)) * (hd ** -0.5 * LOG2E)).to(BF)
    k = unit_rms(torch.randn(P, Bkv, Lloc, D, device=DEV, generator=g)).to(BF)
    v = torch.randn(P, Bkv, Lloc, D, device=DEV, generator=g).to(BF)
    return q, k, v, g


def _bound(q, k, H, hd):
    """the sufficient condition the model's own bound comes from (Cauchy-Schwarz per head), on the data (log2 units)"""
    qn = q.float().view(*q.shape[:-1], H, hd).norm(dim=-1).amax().item()
    kn = k.float().view(*k.shape[:-1], H, hd).norm(dim=-1).amax().item()
    return 1.02 * qn * kn


def _ref_rows(q, k, v, rows, H, hd, v_deq=None):
    """fp64 softmax(q k^T) v for sample (batch, row) pairs; q is pre-scaled (log2 units): softmax in base 2"""
    P, Bkv, Lloc, D = k.shape
    outs, lses = [], []
    for b, r in rows:
        bk = b % Bkv
        qh = q[b, r].double().view(H, 1, hd)
        kh = k[:, bk].reshape(P * Lloc, H, hd).double().permute(1, 2, 0)            # [H, hd, L]
        s2 = qh @ kh                                                                # [H, 1, L], log2 units
        p_ = torch.softmax(s2 * (1.0 / LOG2E), -1)
        vv = (v if v_deq is None else v_deq)[:, bk].reshape(P * Lloc, H, hd).double().permute(1, 0, 2)
        outs.append((p_ @ vv).reshape(D))
        lses.append(torch.logsumexp(s2 * (1.0 / LOG2E), -1).reshape(H))
    return torch.stack(outs), torch.stack(lses)


def _sample_rows(Bq, Lq, rows_per_unit):
    nu = (Lq + rows_per_unit - 1) // rows_per_unit
    rs = sorted({0, 31, rows_per_unit - 1, rows_per_unit, (nu - 1) * rows_per_unit - 1, (nu - 1) * rows_per_unit, Lq - 1})
    return [(b, r) for b in sorted({0, Bq - 1}) for r in rs if 0 <= r < Lq]


CASES = [
    # name,                 hd,  H, P, Lloc,  Bq, Bkv, Lq
    ("cfg3_xl_allgather",   72, 16, 8, 23014, 1,  0,  2700),
    ("cfg3_xl_heads",       72,  2, 8, 23014, 8,  1,  2700),
    ("cfg4_11b_allgather", 128,  6, 8, 28864, 1,  0,  1500),
    ("cfg4_11b_heads",     128,  3, 8, 28864, 8,  1,  1500),
    ("768px_11b_heads",    128,  3, 8,  9568, 24, 3,  1200),
    ("768px_xl_heads",      72,  2, 8,  9568, 24, 3,  2100),
]


@pytest.mark.parametrize("name,hd,H,P,Lloc,Bq,Bkv,Lq", CASES, ids=[c[0] for c in CASES])
def test_rank_attention_bounded_body(hip_lib, name, hd, H, P, Lloc, Bq, Bkv, Lq):
    """the FAST (bounded) body at a rank's key layout -- for head_dim 72 in BOTH unit layouts (256-row and the wide 512-row
    one the rank's launch picks when it fills the chip)"""
    D = H * hd
    kvb = Bkv if Bkv else Bq
    q, k, v, g = _qk(P, Bq, kvb, Lq, Lloc, H, hd, seed=91)
    bound = _bound(q, k, H, hd)
    assert bound <= 56.0
    assert "FAST" in hip_lib.attention_body(hd, P, Lloc, bound)
    segp = (Lloc + 63) // 64 * 64
    vts = torch.zeros(P, kvb, H, hd, segp, dtype=BF, device=DEV)
    hip_lib.v_transpose(v.view(P * kvb, Lloc, D), vts.view(P * kvb, H, hd, segp), H, hd)
    ws = hip_lib.attention_workspace(q.device)
    layouts = (256, 512) if hd == 72 else (256,)
    outs = {}
    try:
        for rows in layouts:
            hip_lib.lib.osk_attention_rows_override(rows)
            out = torch.empty(Bq, Lq, D, dtype=BF, device=DEV)
            lse = torch.empty(Bq, H, Lq, dtype=torch.float32, device=DEV)
            hip_lib.attention_fwd(q, k[0], vts, out, H, hd, hd ** -0.5, lse=lse, n_seg=P, seg_len=Lloc,
                                  k_seg_stride=k.stride(0), vt_seg_stride=vts.stride(0), q_prescaled=True, kv_batches=Bkv,
                                  workspace=ws, score_bound=bound)
            sample = _sample_rows(Bq, Lq, rows)
            ref, ref_lse = _ref_rows(q, k, v, sample, H, hd)
            got = torch.stack([out[b, r].double() for b, r in sample])
            got_lse = torch.stack([lse[b, :, r].double() for b, r in sample])
            assert (got - ref).abs().max().item() <= 2.5e-2, (name, rows)
            assert (got_lse - ref_lse).abs().max().item() <= 4e-3, (name, rows)
            outs[rows] = out
    finally:
        hip_lib.lib.osk_attention_rows_override(0)
    if len(outs) == 2:   # the two layouts order the same f32 sums differently inside a tile only: agree to one bf16 step
        assert (outs[256].float() - outs[512].float()).abs().max().item() <= 2 ** -7 * outs[256].float().abs().max().item() + 1e-3
    # the launch the rank's call makes by itself (no override) is one of the two above
    parts, rows = hip_lib.attention_launch_shape(Bq, H, Lq, P, Lloc, hd, bound, ws.numel())
    assert rows in layouts and 1 <= parts <= 8
    # constant V: softmax rows sum to one over all P x Lloc keys, for every query row
    c = torch.randn(D, device=DEV, generator=g).to(BF)
    hip_lib.v_transpose(c[None, None].expand(P * kvb, Lloc, D).contiguous(), vts.view(P * kvb, H, hd, segp), H, hd)
    out = torch.empty(Bq, Lq, D, dtype=BF, device=DEV)
    hip_lib.attention_fwd(q, k[0], vts, out, H, hd, hd ** -0.5, n_seg=P, seg_len=Lloc, k_seg_stride=k.stride(0),
                          vt_seg_stride=vts.stride(0), q_prescaled=True, kv_batches=Bkv, workspace=ws, score_bound=bound)
    assert (out.float() - c.float()[None, None]).abs().max().item() <= 2 ** -6 * c.float().abs().max().item() + 1e-3


PV8_CASES = [c for c in CASES if c[0] in ("cfg4_11b_allgather", "cfg4_11b_heads", "768px_11b_heads", "cfg3_xl_heads")]


@pytest.mark.parametrize("name,hd,H,P,Lloc,Bq,Bkv,Lq", PV8_CASES, ids=[c[0] for c in PV8_CASES])
def test_rank_attention_fp8_pv_body(hip_lib, name, hd, H, P, Lloc, Bq, Bkv, Lq):
    """BASELINE configs[4] ("fp8 MFMA"): the fp8 P.V body over 8 x 28,864 keys (longest tested key axis so far: 4,096)"""
    D = H * hd
    kvb = Bkv if Bkv else Bq
    q, k, v, g = _qk(P, Bq, kvb, Lq, Lloc, H, hd, seed=93)
    segp = (Lloc + 63) // 64 * 64
    RP = hip_lib.vt8_rows(hd)
    sv = (v.float().abs().view(P, kvb, Lloc, H, hd).amax(dim=(0, 2, 4)) / 448.0).contiguous()        # [kvb, H]: one scale over all segments
    vt8 = torch.zeros(P, kvb, H, RP, segp, dtype=torch.uint8, device=DEV)
    for s_ in range(P):
        hip_lib.v_transpose_fp8(v[s_], sv, vt8[s_], H, hd)
    ws = hip_lib.attention_workspace(q.device)
    out = torch.empty(Bq, Lq, D, dtype=BF, device=DEV)
    lse = torch.empty(Bq, H, Lq, dtype=torch.float32, device=DEV)
    hip_lib.attention_fwd_pv8(q, k[0], vt8, sv, out, H, hd, hd ** -0.5, lse=lse, n_seg=P, seg_len=Lloc, k_seg_stride=k.stride(0),
                              vt_seg_stride=vt8.stride(0), q_prescaled=True, kv_batches=Bkv, workspace=ws)
    sample = _sample_rows(Bq, Lq, 256)
    vf = v.float().view(P, kvb, Lloc, H, hd)
    s4 = sv[None, :, None, :, None]
    v_deq = ((vf / s4).clamp(-448, 448).to(F8).float() * s4).view(P, kvb, Lloc, D)
    ref8, ref_lse = _ref_rows(q, k, v, sample, H, hd, v_deq=v_deq)         # exact P, the kernel's e4m3 V
    ref, _ = _ref_rows(q, k, v, sample, H, hd)
    got = torch.stack([out[b, r].double() for b, r in sample])
    got_lse = torch.stack([lse[b, :, r].double() for b, r in sample])
    rel8 = ((got - ref8).norm() / ref8.norm()).item()
    rel = ((got - ref).norm() / ref.norm()).item()
    # the output of a 200k-key softmax of near-uniform weights is an average of ~N(0,1) values: |out| ~ L^-1/2, so the
    # e4m3 noise of P and V (3 % per element, averaged the same way) stays a fixed FRACTION of it -- same gate as tests/test_gpu_fp8.py
    assert rel8 <= 4e-2 and rel <= 6e-2, (name, rel8, rel)
    assert (got_lse - ref_lse).abs().max().item() <= 7e-2, name
    # constant V: out == e4m3(c / s) s exactly (the denominator is the ones row of the same fp8 product)
    c = torch.randn(D, device=DEV, generator=g).to(BF)
    vc = c[None, None].expand(kvb, Lloc, D).contiguous()
    svc = (vc.float().abs().view(kvb, Lloc, H, hd).amax(dim=(1, 3)) / 448.0).contiguous()
    for s_ in range(P):
        hip_lib.v_transpose_fp8(vc, svc, vt8[s_], H, hd)
    hip_lib.attention_fwd_pv8(q, k[0], vt8, svc, out, H, hd, hd ** -0.5, n_seg=P, seg_len=Lloc, k_seg_stride=k.stride(0),
                              vt_seg_stride=vt8.stride(0), q_prescaled=True, kv_batches=Bkv, workspace=ws)
    c8 = (c.float().view(H, hd) / svc[0][:, None]).clamp(-448, 448).to(F8).float() * svc[0][:, None]
    assert (out.float() - c8.view(1, 1, D)).abs().max().item() <= 2 ** -7 * c8.abs().max().item() + 1e-3
