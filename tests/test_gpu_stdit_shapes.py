"""STDiT-style attention call shapes on the MMDiT kernels (BASELINE.json's north_star vocabulary).

The mounted reference (Open-Sora v2.0) contains no STDiT source, so there is nothing to pin these against except the
mathematics: **parity unpinned**, fp64 softmax(QK^T / sqrt(d)) V as the oracle.  What is shown: the v1.x block's three
attention patterns are plain views / call shapes of osk_attention_fwd_bf16 at STDiT-XL/2 geometry (hidden 1152, 16 heads,
head_dim 72 -> the hand-scheduled kernel):
  spatial  self-attention over H*W : batch = B*T,   sequence = H*W
  temporal self-attention over T   : batch = B*H*W, sequence = T        (one ragged 64-key tile)
  cross attention to the T5 tokens : q_len = T*H*W, kv_len = 512 (model_max_length), no RoPE
Round 4 (VERDICT r3 "missing" 1): the two remaining pieces of that vocabulary, both parity-unpinned for the same reason --
  * the GEGLU up-projection as ONE GEMM whose epilogue multiplies value and gelu(gate) in registers (osk_gemm_geglu_bf16),
  * the temporal attention as its own HBM-bound kernel, one wave per (sequence, head), with flash-attn's `alibi_slopes`
    semantics (osk_attention_short_bf16) -- against fp64 oracles written here."""
import pytest
import torch

from tests.test_gpu_kernels import rnd

pytestmark = pytest.mark.gpu
DEV, BF = "cuda", torch.bfloat16
H, HD = 16, 72
D = H * HD


def _ref(q, k, v):
    """q [N, Lq, D], k / v [N, Lk, D] -> fp64 attention per head"""
    N, Lq, _ = q.shape
    Lk = k.shape[1]
    qh = q.float().cpu().view(N, Lq, H, HD).permute(0, 2, 1, 3).double()
    kh = k.float().cpu().view(N, Lk, H, HD).permute(0, 2, 1, 3).double()
    vh = v.float().cpu().view(N, Lk, H, HD).permute(0, 2, 1, 3).double()
    return (torch.softmax(qh @ kh.transpose(-1, -2) * HD ** -0.5, -1) @ vh).permute(0, 2, 1, 3).reshape(N, Lq, D)


def _attn(hip_lib, q, k, v):
    N, Lq, _ = q.shape
    Lk = k.shape[1]
    vt = torch.empty(N, H, HD, (Lk + 63) // 64 * 64, dtype=BF, device=DEV)
    hip_lib.v_transpose(v, vt, H, HD)
    out = torch.empty(N, Lq, D, dtype=BF, device=DEV)
    hip_lib.attention_fwd(q, k, vt, out, H, HD, HD ** -0.5)
    return out


def test_spatial_temporal_and_cross_attention_shapes(hip_lib):
    B, T, Hh, Ww, Ltxt = 1, 6, 12, 16, 512
    S = Hh * Ww
    x = rnd("x", (B, T, S, 3 * D), seed=61)          # [B, T, H*W, (q k v)] tokens of one STDiT block input
    q, k, v = x[..., :D], x[..., D: 2 * D], x[..., 2 * D:]
    # spatial: (B T) batches of H*W tokens -- a reshape
    qs, ks, vs = (t.reshape(B * T, S, D) for t in (q, k, v))
    got = _attn(hip_lib, qs, ks, vs)
    assert (got.float().cpu().double() - _ref(qs, ks, vs)).abs().max().item() <= 2.5e-2
    # temporal: (B H*W) batches of T tokens -- a transpose view made contiguous once
    qt, kt, vt_ = (t.permute(0, 2, 1, 3).reshape(B * S, T, D).contiguous() for t in (q, k, v))
    got = _attn(hip_lib, qt, kt, vt_)
    assert (got.float().cpu().double() - _ref(qt, kt, vt_)).abs().max().item() <= 2.5e-2
    # cross: all T*H*W image tokens against 512 text tokens (kv from another tensor, q_len != kv_len)
    y = rnd("y", (B, Ltxt, 2 * D), seed=62)
    qc = q.reshape(B, T * S, D)
    got = _attn(hip_lib, qc, y[..., :D], y[..., D:])
    assert (got.float().cpu().double() - _ref(qc, y[..., :D], y[..., D:])).abs().max().item() <= 2.5e-2


# ------------------------------------------------------------------------------------------------ GEGLU up-projection
def _geglu_ref(a, wv, wg, bv, bg):
    x = a.double().cpu()
    v = x @ wv.double().cpu().T + bv.double().cpu()
    g = x @ wg.double().cpu().T + bg.double().cpu()
    return v * torch.nn.functional.gelu(g.float(), approximate="tanh").double()


@pytest.mark.parametrize("B,L,N_out,K,path", [
    (2, 3072, 4608, 1152, "fused"),     # STDiT-XL/2 MLP width (4 x 1152) at a spatial block's token count: the 256 x 256 tile kernel
    (3, 2000, 1024, 576, "fused"),      # ragged M (23.4 row tiles: edge tiles in M)
    (2, 3000, 1056, 256, "fused"),      # 2 N_out = 2112 = 8.25 column tiles: the last tile is a ragged edge in N
    (1, 77, 384, 128, "fallback"),      # small: plain GEMM into the workspace + the row kernel
])
def test_geglu_up_projection_vs_f64(hip_lib, B, L, N_out, K, path):
    a = rnd("a", (B, L, K), seed=91)
    wv, wg = rnd("wv", (N_out, K), std=K ** -0.5, seed=92), rnd("wg", (N_out, K), std=K ** -0.5, seed=93)
    bv = rnd("bv", (N_out,), std=0.2, dtype=torch.float32, seed=94)
    bg = rnd("bg", (N_out,), std=0.2, dtype=torch.float32, seed=95)
    wp, bp = hip_lib.geglu_pack(wv, wg, bv, bg)
    assert torch.equal(wp[:16], wv[:16]) and torch.equal(wp[16:32], wg[:16]) and torch.equal(wp[32:48], wv[16:32])
    out = torch.full((B, L, N_out), float("nan"), dtype=BF, device=DEV)
    ws = torch.empty(B * L * 2 * N_out, dtype=BF, device=DEV)
    if path == "fallback":
        with pytest.raises(RuntimeError):
            hip_lib.gemm_geglu(a, wp, bp, out)                 # no workspace: nothing is launched
    hip_lib.gemm_geglu(a, wp, bp, out, workspace=None if path == "fused" else ws)
    ref = _geglu_ref(a, wv, wg, bv, bg)
    got = out.float().cpu().double()
    assert torch.isfinite(got).all()
    # one bf16 rounding of the product (the fallback rounds value and gate to bf16 first: a second half-ulp)
    tol = 2 ** -7 if path == "fused" else 2 ** -6
    assert ((got - ref).abs() <= tol * ref.abs() + 3e-3).all(), ((got - ref).abs().max().item(), path)
    # strided output view (a column slice of a wider buffer) and no bias
    wide = torch.zeros(B, L, N_out + 64, dtype=BF, device=DEV)
    wp0, _ = hip_lib.geglu_pack(wv, wg)
    hip_lib.gemm_geglu(a, wp0, None, wide[:, :, 32:32 + N_out], workspace=ws)
    ref0 = _geglu_ref(a, wv, wg, torch.zeros_like(bv), torch.zeros_like(bg))
    assert ((wide[:, :, 32:32 + N_out].float().cpu().double() - ref0).abs() <= 2 ** -6 * ref0.abs() + 3e-3).all()
    assert float(wide[:, :, :32].abs().max()) == 0.0 and float(wide[:, :, 32 + N_out:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ temporal attention + ALiBi
def _alibi_ref(q, k, v, Hh, hd, slopes):
    N, Lq, _ = q.shape
    Lk = k.shape[1]
    qh = q.float().cpu().view(N, Lq, Hh, hd).permute(0, 2, 1, 3).double()
    kh = k.float().cpu().view(N, Lk, Hh, hd).permute(0, 2, 1, 3).double()
    vh = v.float().cpu().view(N, Lk, Hh, hd).permute(0, 2, 1, 3).double()
    s = qh @ kh.transpose(-1, -2) * hd ** -0.5
    if slopes is not None:
        i = torch.arange(Lq, dtype=torch.float64)[:, None]
        j = torch.arange(Lk, dtype=torch.float64)[None, :]
        s = s - slopes.double().cpu()[None, :, None, None] * (i + Lk - Lq - j).abs()
    return (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3).reshape(N, Lq, Hh * hd)


@pytest.mark.parametrize("hd,Hh", [(72, 16), (64, 6), (128, 4)])
@pytest.mark.parametrize("Lq,Lk", [(16, 16), (51, 51), (64, 64), (7, 33), (33, 64), (1, 1)])
@pytest.mark.parametrize("alibi", [False, True])
def test_temporal_attention_short_kernel_with_alibi_vs_f64(hip_lib, hd, Hh, Lq, Lk, alibi):
    """STDiT temporal self-attention call shape (B * H * W sequences of T frames) on the one-wave-per-unit kernel; ALiBi slopes as
    in the ALiBi paper (2^(-8 h / H)); also cross-length (Lq != Lk: flash-attn's bottom-right aligned distance)."""
    N = 150
    Dm = Hh * hd
    q, k, v = rnd("q", (N, Lq, Dm), seed=71), rnd("k", (N, Lk, Dm), seed=72), rnd("v", (N, Lk, Dm), seed=73)
    slopes = torch.tensor([2.0 ** (-8.0 * (h + 1) / Hh) for h in range(Hh)], dtype=torch.float32, device=DEV) if alibi else None
    out = torch.full((N, Lq, Dm), float("nan"), dtype=BF, device=DEV)
    hip_lib.attention_short(q, k, v, out, Hh, hd, hd ** -0.5, slopes)
    ref = _alibi_ref(q, k, v, Hh, hd, slopes)
    got = out.float().cpu().double()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 2.5e-2 and ((got - ref).norm() / ref.norm()).item() <= 6e-3
    if alibi and Lk > 8:      # the bias really acts: the unbiased result is far from the biased oracle
        hip_lib.attention_short(q, k, v, out, Hh, hd, hd ** -0.5, None)
        assert (out.float().cpu().double() - ref).abs().max().item() > 5e-2


def test_temporal_attention_short_kernel_on_block_views_matches_flash_kernel(hip_lib):
    """the temporal call shape of test_spatial_temporal_and_cross_attention_shapes on the short kernel: strided views of the block's
    [B, T, H*W, 3 D] projection buffer (no .contiguous() copy of q / k / v), equal to the flash kernel's result to P's bf16 rounding"""
    B, T, S = 1, 16, 12 * 16
    x = rnd("x", (B, T, S, 3 * D), seed=61)
    xt = x.permute(0, 2, 1, 3).reshape(B * S, T, 3 * D).contiguous()      # one transpose of the whole projection row
    q, k, v = xt[..., :D], xt[..., D:2 * D], xt[..., 2 * D:]              # views: row stride 3 D
    out = torch.empty(B * S, T, D, dtype=BF, device=DEV)
    hip_lib.attention_short(q, k, v, out, H, HD, HD ** -0.5)
    ref = _ref(q, k, v)
    assert (out.float().cpu().double() - ref).abs().max().item() <= 2.5e-2
    flash = _attn(hip_lib, q.contiguous(), k.contiguous(), v.contiguous())
    assert ((out.float() - flash.float()).norm() / flash.float().norm()).item() <= 6e-3
    with pytest.raises(RuntimeError):
        hip_lib.attention_short(torch.empty(1, 65, D, dtype=BF, device=DEV), k[:1], v[:1], torch.empty(1, 65, D, dtype=BF, device=DEV), H, HD, 1.0)
