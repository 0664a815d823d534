"""STDiT-style attention call shapes on the MMDiT kernels (BASELINE.json's north_star vocabulary).

The mounted reference (Open-Sora v2.0) contains no STDiT source, so there is nothing to pin these against except the
mathematics: **parity unpinned**, fp64 softmax(QK^T / sqrt(d)) V as the oracle.  What is shown: the v1.x block's three
attention patterns are plain views / call shapes of osk_attention_fwd_bf16 at STDiT-XL/2 geometry (hidden 1152, 16 heads,
head_dim 72 -> the hand-scheduled kernel):
  spatial  self-attention over H*W : batch = B*T,   sequence = H*W
  temporal self-attention over T   : batch = B*H*W, sequence = T        (one ragged 64-key tile)
  cross attention to the T5 tokens : q_len = T*H*W, kv_len = 512 (model_max_length), no RoPE
Not built: the GEGLU GEMM epilogue and ALiBi bias of those blocks."""
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
