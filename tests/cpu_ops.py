"""TEST INFRASTRUCTURE ONLY — a CPU emulation of the *semantics* of every kernel in include/osk.h, with the same
Python call signatures as open_sora_amd/_C.py.  It lets the not-gpu suite run the host-side orchestration
(open_sora_amd/mmdit.py, seqpar.py, sampling.py: buffer views, column offsets, token sharding, collectives under
gloo) against the goldens without a GPU.  It is never imported by the product path and is not a fallback:
open_sora_amd binds its kernel table to the HIP library at import.

Math is fp32 on the bf16-stored operands, outputs rounded once (what the HIP kernels do)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from oracle import mmdit_oracle as O

BF = torch.bfloat16
PROFILE_ATTENTION = None
_PERM = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15])
_PV16_KEYS = torch.tensor([[0, 1, 2, 3, 8, 9, 10, 11], [16, 17, 18, 19, 24, 25, 26, 27], [4, 5, 6, 7, 12, 13, 14, 15], [20, 21, 22, 23, 28, 29, 30, 31]])


def pos2key(hd: int, Lp: int) -> torch.Tensor:
    """key stored at position p of a V^T row (include/osk.h, osk_v_transpose_bf16): the order the attention kernel of that head_dim
    holds its P operand in.  head_dim 72 (16x16x32 P.V): 16-byte chunk c of a 64-key tile = the 8 keys of lane row c % 4 in the
    32-key half c / 4 -- and head_dim 64, which runs the same loop; head_dim 128 (32x32x16): a 4-key swap inside every 16 keys."""
    p = torch.arange(Lp)
    if hd in (64, 72):
        c, e = (p % 64) // 8, p % 8
        return (p // 64) * 64 + 32 * (c // 4) + _PV16_KEYS[c % 4, e]
    return (p // 16 * 16) + _PERM[p % 16]


def _abi_check(what: str, *conds) -> None:
    """the argument checks of the C ABI entry points (OSK_EINVAL conditions of csrc/*.hip), so that host logic
    exercised on the CPU cannot hand the kernels a view they would refuse on the GPU"""
    if not all(conds):
        raise RuntimeError(f"{what} failed: status -1 (invalid argument / unsupported shape)")


def _al(t, nbytes: int) -> bool:
    """data pointer alignment of a view whose storage base is allocator-aligned"""
    return t is None or (t.storage_offset() * t.element_size()) % nbytes == 0


def _mod_rows(t: torch.Tensor, batch_stride: int, B: int, D: int) -> torch.Tensor:
    """f32 view whose row b starts batch_stride elements after row b-1 (pointer-carrier convention of _C)."""
    if B == 1:
        return t.reshape(-1)[:D][None]
    assert t.stride(0) == batch_stride or t.shape[0] == 1
    return t[:, :D]


def ln_modulate(x, shift, scale, out, mod_batch_stride, eps=1e-6):
    B, L, D = x.shape
    y = F.layer_norm(x.float(), (D,), eps=eps)
    out.copy_(((1 + _mod_rows(scale, mod_batch_stride, B, D)[:, None]) * y + _mod_rows(shift, mod_batch_stride, B, D)[:, None]).to(out.dtype))
    return out


def gemm(a, w, bias, out, *, res=None, gate=None, gate_batch_stride=0, gelu_from=None):
    B, L, K = a.shape
    N = w.shape[0]
    assert K % 64 == 0, "osk_gemm_bf16 requires K % 64 == 0"
    _abi_check("osk_gemm_bf16", a.stride(2) == 1, a.stride(0) % 8 == 0, a.stride(1) % 8 == 0, w.stride(0) % 8 == 0,
               out.stride(0) % 4 == 0, out.stride(1) % 4 == 0, _al(a, 16), _al(w, 16), _al(out, 8), _al(bias, 16),
               gate is None or (res is not None and _al(gate, 16) and gate_batch_stride % 4 == 0 and _al(res, 8)))
    assert res is None or res.stride() == out.stride(), "osk_gemm_bf16: the residual shares C's strides (the ABI carries no residual strides)"
    v = a.float() @ w.float().T
    if bias is not None:
        v = v + bias.float()
    if gelu_from is not None and gelu_from < N:
        v = torch.cat([v[..., :gelu_from], F.gelu(v[..., gelu_from:], approximate="tanh")], -1)
    if gate is not None:
        v = res.float() + _mod_rows(gate, gate_batch_stride, B, N)[:, None] * v
    out.copy_(v.to(out.dtype))
    return out


def gemm_pair(first, second, *, gelu_from=None):
    """osk_gemm_bf16_pair == the two osk_gemm_bf16 calls"""
    assert tuple(first["w"].shape) == tuple(second["w"].shape)
    for d in (first, second):
        gemm(d["a"], d["w"], d["bias"], d["out"], res=d.get("res"), gate=d.get("gate"), gate_batch_stride=d.get("gate_batch_stride", 0),
             gelu_from=gelu_from)


def gemm_group(tasks):
    """CPU statement of osk_gemm_group_bf16 (include/osk.h): plain tasks = osk_gemm_bf16 without the skipped physical columns; V^T
    tasks = the projection rounded once to bf16, written key-major in osk_v_transpose_bf16's order behind `vt_pos`.  Returns False
    (nothing done) for groups the library declines: a task off the 256 x 256 tile path."""
    for d in tasks:
        if "vt" in d:
            B, L, K = d["x"].shape
            if d["w"].shape[0] < 256 or B * ((L + 63) // 64 * 64) < 128 or d["hd"] not in (64, 72, 128):
                return False
        else:
            B, L, K = d["a"].shape
            sl = d.get("skip", (0, 0))[1]
            if B * L < 256 or d["w"].shape[0] - sl < 128:
                return False
            if sl and (d["skip"][0] % 256 or sl % 8 or d.get("gate") is not None):
                raise RuntimeError("osk_gemm_group_bf16 failed: status -1 (invalid skip range)")
    for d in tasks:
        if "vt" in d:
            x, w, vt, hd = d["x"], d["w"], d["vt"], d["hd"]
            B, L, K = x.shape
            _abi_check("osk_gemm_group_bf16 (V^T task)", x.stride(2) == 1, x.stride(0) % 8 == 0, x.stride(1) % 8 == 0, w.stride(0) % 8 == 0,
                       _al(x, 16), _al(w, 16), d["vt_pos"] % 64 == 0, vt.is_contiguous())
            v = x.float() @ w.float().T
            if d.get("bias") is not None:
                v = v + d["bias"].float()
            H = w.shape[0] // hd
            Lp = (L + 63) // 64 * 64
            part = torch.zeros(B, H, hd, Lp, dtype=torch.bfloat16)
            v_transpose(v.to(torch.bfloat16), part, H, hd)
            vt[..., d["vt_pos"]: d["vt_pos"] + Lp] = part
        else:
            a, w, out = d["a"], d["w"], d["out"]
            sf, sl = d.get("skip", (0, 0))
            N = w.shape[0]
            gf = d.get("gelu_from")
            bias = d.get("bias")
            for lo, hi in ((0, sf), (sf + sl, N)) if sl else ((0, N),):
                if hi <= lo:
                    continue
                g = None if gf is None or gf >= hi else max(gf - lo, 0)
                res = d.get("res")
                gate = d.get("gate")
                gemm(a, w[lo:hi], None if bias is None else bias[lo:hi], out[:, :, lo:hi], res=None if res is None else res[:, :, lo:hi],
                     gate=None if gate is None else gate[:, lo:hi], gate_batch_stride=d.get("gate_batch_stride", 0), gelu_from=g)
    return True


def ln_modulate_fp8(x, shift, scale, mod_batch_stride, eps=1e-6):
    """osk_ln_modulate_fp8 == osk_ln_modulate_bf16 followed by osk_quantize_rows_fp8 (bit-identical by construction)"""
    xm = torch.empty(x.shape, dtype=torch.bfloat16)
    ln_modulate(x, shift, scale, xm, mod_batch_stride, eps)
    return quantize_rows_fp8(xm)


def quantize_rows_fp8(x, out8=None, scales=None):
    """CPU statement of osk_quantize_rows_fp8 (torch's own float8_e4m3fn conversion)"""
    if x.dim() == 2:
        x = x.unsqueeze(0)
    B, L, K = x.shape
    xf = x.float().reshape(B * L, K)
    amax = xf.abs().amax(-1)
    inv = torch.where(amax > 0, torch.tensor(448.0) / amax, torch.zeros_like(amax))
    q = (xf * inv[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    return q, torch.where(amax > 0, amax / torch.tensor(448.0), torch.ones_like(amax))


def gemm_fp8_supported(M, N, K):
    return M >= 256 and N >= 128 and K % 128 == 0


def gemm_fp8(a8, a_scale, w8, w_scale, bias, out, *, res=None, gate=None, gate_batch_stride=0, gelu_from=None):
    M, K = a8.shape
    B, L, N = out.shape
    assert gemm_fp8_supported(M, N, K), "osk_gemm_fp8 refuses this shape"
    a = (a8.view(torch.float8_e4m3fn).float() * a_scale[:, None]).view(B, L, K)
    w = w8.view(torch.float8_e4m3fn).float() * w_scale[:, None]
    v = a @ w.T
    if bias is not None:
        v = v + bias.float()
    if gelu_from is not None and gelu_from < N:
        v = torch.cat([v[..., :gelu_from], F.gelu(v[..., gelu_from:], approximate="tanh")], -1)
    if gate is not None:
        v = res.float() + _mod_rows(gate, gate_batch_stride, B, N)[:, None] * v
    out.copy_(v.to(out.dtype))
    return out


class GemvTasks:
    def __init__(self, layers, device):
        self.layers = layers


def gemv_tasks(x, tasks, out, act_in=0, accumulate=False):
    xx = F.silu(x.float()) if act_in == 1 else x.float()
    for w, b, col in tasks.layers:
        r = xx @ w.float().T
        if b is not None:
            r = r + b.float()
        sl = out[:, col: col + w.shape[0]]
        sl.copy_(sl + r if accumulate else r)
    return out


def timestep_embedding(t, out, max_period=10000.0, time_factor=1000.0):
    out.copy_(O.timestep_embedding(t.float(), out.shape[1], max_period, time_factor))
    return out


def rope_table(ids, axes_dim, theta, f32_angles, cos, sin):
    ang = (O.rope_angles_liger if f32_angles else O.rope_angles)(ids[None], axes_dim, theta)[0]
    cos.view(-1, cos.shape[-1]).copy_(torch.cos(ang).float())
    sin.view(-1, sin.shape[-1]).copy_(torch.sin(ang).float())


def qknorm_rope(q, k, qs0, ks0, qs1, ks1, l_split, cos, sin, cs_batch_stride, H, hd, rope_mode, eps=1e-6, q_mult=1.0):
    B, L, _ = (q if q is not None else k).shape
    c = cos if cs_batch_stride else cos[:1]
    s = sin if cs_batch_stride else sin[:1]
    for t, (s0, s1) in ((q, (qs0, qs1)), (k, (ks0, ks1))):
        if t is None:
            continue
        x = t.reshape(B, L, H, hd)
        y = torch.cat([O.rms_norm(x[:, :l_split], s0), O.rms_norm(x[:, l_split:], s1)], 1)  # bf16 rounding points
        yf = y.float()
        cc, ss = c[:, :, None, :], s[:, :, None, :]
        if rope_mode == 0:
            p = yf.reshape(B, L, H, hd // 2, 2)
            o = torch.stack([cc * p[..., 0] - ss * p[..., 1], ss * p[..., 0] + cc * p[..., 1]], -1).reshape(B, L, H, hd)
        else:
            x1, x2 = yf[..., : hd // 2], yf[..., hd // 2:]
            o = torch.cat([x1 * cc - x2 * ss, x2 * cc + x1 * ss], -1)
        if t is q:
            o = o * q_mult  # folded into q before its single rounding (kernel semantics)
        t.copy_(o.reshape(B, L, H * hd).to(t.dtype))


def v_transpose(v, vt, H, hd):
    B, L, _ = v.shape
    Lp = vt.shape[-1]
    pad = torch.zeros(B, Lp, H, hd, dtype=v.dtype)
    pad[:, :L] = v.reshape(B, L, H, hd)
    vt.copy_(pad[:, pos2key(hd, Lp)].permute(0, 2, 3, 1))


def vt8_rows(hd):
    return (hd + 1 + 15) // 16 * 16


def v_scale_fp8(v, H, hd):
    B, L, _ = v.shape
    amax = v.float().reshape(B, L, H, hd).abs().amax(dim=(1, 3))
    return torch.where(amax > 0, amax / 448.0, torch.ones_like(amax)).contiguous()


def v_transpose_fp8(v, scales, vt8, H, hd):
    """CPU statement of osk_v_transpose_fp8.  The key order inside a 64-key tile is the kernel's private business (it only
    has to agree with attention_fwd_pv8); this emulation keeps the natural order."""
    B, L, _ = v.shape
    Lp = vt8.shape[-1]
    _abi_check("osk_v_transpose_fp8", v.stride(0) % 8 == 0, v.stride(1) % 8 == 0, _al(v, 16), _al(vt8, 16),
               vt8.shape[-2] == vt8_rows(hd), Lp == (L + 63) // 64 * 64, scales.numel() == B * H)
    x = v.float().reshape(B, L, H, hd) / scales.reshape(B, 1, H, 1)
    q = x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8).permute(0, 2, 3, 1)        # [B, H, hd, L]
    vt8.zero_()
    vt8[:, :, :hd, :L] = q
    vt8[:, :, hd, :L] = 0x38                                                                          # 1.0 in e4m3
    return vt8


def attention_fwd_pv8(q, k, vt8, v_scale, out, H, hd, scale, *, lse=None, n_seg=1, seg_len=None, k_seg_stride=0,
                      vt_seg_stride=0, q_prescaled=False, kv_batches=0, workspace=None):
    """CPU statement of osk_attention_fwd_pv8_bf16: exact QK^T and softmax bookkeeping, P and V as e4m3, the
    denominator = the sum of the SAME e4m3 P (ones row of V^T)."""
    Bq, Lq, D = q.shape
    B = kv_batches if kv_batches else Bq
    if seg_len is None:
        seg_len = k.shape[1]
    seg_lp = (seg_len + 63) // 64 * 64
    RP = vt8_rows(hd)
    _abi_check("osk_attention_fwd_pv8_bf16", q.stride(0) % 8 == 0, q.stride(1) % 8 == 0, k.stride(0) % 8 == 0,
               k.stride(1) % 8 == 0, k_seg_stride % 8 == 0, vt_seg_stride % 16 == 0, out.stride(0) % 4 == 0,
               out.stride(1) % 4 == 0, _al(q, 16), _al(k, 16), _al(vt8, 16), _al(out, 8), 0 <= kv_batches <= Bq,
               v_scale.numel() == B * H)
    ks, vs, ones = [], [], []
    for s in range(n_seg):
        k_s = k if n_seg == 1 else torch.as_strided(k, (B, seg_len, D), k.stride(), k.storage_offset() + s * k_seg_stride)
        vt_s = torch.as_strided(vt8, (B, H, RP, seg_lp), (H * RP * seg_lp, RP * seg_lp, seg_lp, 1),
                                vt8.storage_offset() + s * vt_seg_stride)
        ks.append(k_s.float().reshape(B, seg_len, H, hd))
        f = vt_s.view(torch.float8_e4m3fn).float()
        vs.append(f[:, :, :hd, :seg_len].permute(0, 3, 1, 2))                                       # [B, seg, H, hd]
        ones.append(f[:, :, hd, :seg_len])                                                           # [B, H, seg]
    K = torch.cat(ks, 1).permute(0, 2, 1, 3)
    V = torch.cat(vs, 1).permute(0, 2, 1, 3) * v_scale.reshape(B, H, 1, 1)
    valid = torch.cat(ones, 2)                                                                       # [B, H, Lk]
    Q = q.float().reshape(Bq, Lq, H, hd).permute(0, 2, 1, 3)
    if Bq != B:
        idx = torch.arange(Bq) % B
        K, V, valid = K[idx], V[idx], valid[idx]
    s_ = (Q @ K.transpose(-1, -2)) * (1.0 if q_prescaled else scale * 1.4426950408889634)            # log2 units
    m = s_.max(-1, keepdim=True).values
    P = torch.exp2(s_ - m).to(torch.float8_e4m3fn).float()
    l = (P * valid[:, :, None, :]).sum(-1, keepdim=True)
    o = (P @ V) / l
    out.copy_(o.permute(0, 2, 1, 3).reshape(Bq, Lq, D).to(out.dtype))
    if lse is not None:
        lse.copy_(((m + torch.log2(l)) * 0.6931471805599453).squeeze(-1))
    return out


def attention_workspace(device):
    return torch.empty(16, dtype=torch.uint8, device=device)


def attention_fwd(q, k, vt, out, H, hd, scale, *, lse=None, n_seg=1, seg_len=None, k_seg_stride=0, vt_seg_stride=0,
                  q_prescaled=False, kv_batches=0, workspace=None, score_bound=0.0):
    Bq, Lq, D = q.shape
    B = kv_batches if kv_batches else Bq     # key / value batches; query batch b reads key batch b % B
    _abi_check("osk_attention_fwd_bf16", q.stride(0) % 8 == 0, q.stride(1) % 8 == 0, k.stride(0) % 8 == 0,
               k.stride(1) % 8 == 0, k_seg_stride % 8 == 0, vt_seg_stride % 8 == 0, out.stride(0) % 4 == 0,
               out.stride(1) % 4 == 0, _al(q, 16), _al(k, 16), _al(vt, 16), _al(out, 8), 0 <= kv_batches <= Bq)
    if seg_len is None:
        seg_len = k.shape[1]
    seg_lp = (seg_len + 63) // 64 * 64
    # rebuild the key-major K and V from the segment layout
    ks, vs = [], []
    key2pos = torch.empty(seg_lp, dtype=torch.long)
    key2pos[pos2key(hd, seg_lp)] = torch.arange(seg_lp)
    for s in range(n_seg):
        if n_seg == 1:
            k_s, vt_s = k, vt
        else:
            k_s = torch.as_strided(k, (B, seg_len, D), k.stride(), k.storage_offset() + s * k_seg_stride)
            vt_s = torch.as_strided(vt, (B, H, hd, seg_lp), (H * hd * seg_lp, hd * seg_lp, seg_lp, 1),
                                    vt.storage_offset() + s * vt_seg_stride)
        ks.append(k_s.float().reshape(B, seg_len, H, hd))
        vs.append(vt_s.float().reshape(B, H, hd, seg_lp)[..., key2pos][..., :seg_len].permute(0, 3, 1, 2))
    K = torch.cat(ks, 1).permute(0, 2, 1, 3)
    V = torch.cat(vs, 1).permute(0, 2, 1, 3)
    Q = q.float().reshape(Bq, Lq, H, hd).permute(0, 2, 1, 3)
    if Bq != B:
        idx = torch.arange(Bq) % B
        K, V = K[idx], V[idx]
    s_ = (Q @ K.transpose(-1, -2)) * (0.6931471805599453 if q_prescaled else scale)  # prescaled q: log2 units
    if score_bound:   # the caller's promise (include/osk.h, osk_attention_fwd_bounded_bf16): checked here on every call
        worst = float(s_.abs().max()) / 0.6931471805599453
        assert worst <= score_bound, f"score bound {score_bound} violated: |score| reaches {worst} (log2 units)"
    o = torch.softmax(s_, -1) @ V
    res = o.permute(0, 2, 1, 3).reshape(Bq, Lq, D).to(out.dtype)
    out.copy_(res)
    if lse is not None:
        lse.copy_(torch.logsumexp(s_, -1))
    return out


def rownorm2_max(x, out, H, hd, accumulate=False):
    """CPU statement of osk_rownorm2_max_bf16"""
    B, L, _ = x.shape
    n2 = x.float().reshape(B, L, H, hd).pow(2).sum(-1).amax(1)
    out.copy_(torch.maximum(out, n2) if accumulate else n2)
    return out


def attention_fwd_auto(q, k, vt, out, H, hd, scale, qn2, kn2, *, lse=None, n_seg=1, seg_len=None, k_seg_stride=0, vt_seg_stride=0,
                       q_prescaled=True, kv_batches=0, workspace=None):
    """osk_attention_fwd_auto_bf16: the same softmax whichever body a (batch, head) takes; the emulation checks the norms it was handed
    against the operands (a stale or foreign norm buffer is the host bug this guards)"""
    B = q.shape[0]
    want_q = q.float().reshape(B, q.shape[1], H, hd).pow(2).sum(-1).amax(1)
    assert torch.allclose(qn2.reshape(B, H), want_q, rtol=1e-5), "q_norm2_max does not belong to this q"
    return attention_fwd(q, k, vt, out, H, hd, scale, lse=lse, n_seg=n_seg, seg_len=seg_len, k_seg_stride=k_seg_stride,
                         vt_seg_stride=vt_seg_stride, q_prescaled=q_prescaled, kv_batches=kv_batches, workspace=workspace)


def copy_rows_ok(src, dst):
    if not (src.dtype == dst.dtype == torch.bfloat16 and src.ndim == dst.ndim and src.ndim in (3, 4)):
        return False
    if src.numel() == 0 or src.shape[-1] % 4 or src.stride(-1) != 1 or dst.stride(-1) != 1 or dst.shape[-1] < src.shape[-1]:
        return False
    if any(st % 4 for st in src.stride()[:-1]) or any(st % 4 for st in dst.stride()[:-1]):
        return False
    return src.data_ptr() % 8 == 0 and dst.data_ptr() % 8 == 0


def copy_rows(src, dst):
    assert src.dtype == dst.dtype == torch.bfloat16 and src.ndim == dst.ndim and src.ndim in (3, 4) and src.shape[-1] % 4 == 0
    C = src.shape[-1]
    dst[..., :C].copy_(src.expand(*dst.shape[:-1], C))
    return dst


def cfg_euler(pred, x, x_out, g_txt, g_img, dt, g_img_vec=None):
    c, u, u2 = pred.float().reshape(3, -1)
    gi = g_img if g_img_vec is None else g_img_vec.reshape(-1)
    v = u2 + gi * (u - u2) + g_txt * (c - u)
    x_out.copy_((x.float().reshape(-1) + dt * v).reshape(x.shape).to(x_out.dtype))
    return x_out


# ----------------------------------------------------------------------------------------------
# causal 3-D VAE kernels (NDHWC bf16): semantics of include/osk.h, fp32 math, one rounding
# ----------------------------------------------------------------------------------------------
def conv_out_dims(T, H, W, stride=(1, 1, 1), up=(False, False)):
    Tu = 1 + 2 * (T - 1) if up[0] else T
    Hu, Wu = (2 * H, 2 * W) if up[1] else (H, W)
    return (Tu - 1) // stride[0] + 1, (Hu - 1) // stride[1] + 1, (Wu - 1) // stride[2] + 1


def causal_conv3d(x, w, bias, out, ksize, stride=(1, 1, 1), up=(False, False), res=None, gn_sums=None):
    B, T, H, W, Cin = x.shape
    Cout = w.shape[0]
    taps = ksize ** 3
    wk = w[:, : taps * Cin].float().reshape(Cout, ksize, ksize, ksize, Cin).permute(0, 4, 1, 2, 3)
    assert float(w[:, taps * Cin:].float().abs().sum()) == 0.0, "weight K padding must be zero"
    xs = x.float().permute(0, 4, 1, 2, 3)
    if up[1]:
        xs = xs.repeat_interleave(2, 3).repeat_interleave(2, 4)
    if up[0]:
        xs = torch.cat((xs[:, :, :1], xs[:, :, 1:].repeat_interleave(2, 2)), 2)
    if ksize > 1:
        xs = F.pad(xs, (ksize // 2, ksize // 2, ksize // 2, ksize // 2, ksize - 1, 0), mode="replicate")
    y = F.conv3d(xs, wk, None if bias is None else bias.float(), stride=stride).permute(0, 2, 3, 4, 1)
    if res is not None:
        y = y + res.float()
    out.copy_(y.to(out.dtype))
    if gn_sums is not None:      # the fused-statistics contract: gn_sums arrives zeroed and is ACCUMULATED into
        gn_sums += groupnorm_stats(out, gn_sums.shape[1], torch.empty_like(gn_sums))
        return out, True
    return out


def groupnorm_table(sums, gamma, beta, table, S, G, eps=1e-6):
    B, C = sums.shape[0], gamma.numel()
    n = S * (C // G)
    mean = sums[:, :, 0] / n
    var = (sums[:, :, 1] / n - mean * mean).clamp_min(0)
    rstd = torch.rsqrt(var.float() + eps)
    a = rstd.repeat_interleave(C // G, 1) * gamma.float()[None]
    d = beta.float()[None] - mean.float().repeat_interleave(C // G, 1) * a
    table.copy_(torch.cat((a.reshape(B, C // 8, 8), d.reshape(B, C // 8, 8)), 2))
    return table


def gn_in_supported(x_shape, Cout, ksize, stride):
    """the shapes csrc/conv3d_256.hip::conv256_gn_in_supported takes"""
    B, T, H, W, Cin = x_shape
    return ksize == 3 and tuple(stride) == (1, 1, 1) and H % 16 == 0 and W % 16 == 0 and Cin % 128 == 0 and \
        (Cout >= 256 or (Cout == 128 and T >= 2)) and B * T * H * W >= 256


def causal_conv3d_gn_in(x, table, w, bias, out, ksize, stride=(1, 1, 1), res=None, gn_sums=None):
    B, Cin = x.shape[0], x.shape[-1]
    if not gn_in_supported(x.shape, w.shape[0], ksize, stride):
        return False, False
    a = table[:, :, :8].reshape([B] + [1] * (x.ndim - 2) + [Cin])
    d = table[:, :, 8:].reshape([B] + [1] * (x.ndim - 2) + [Cin])
    y = F.silu((x.float() * a + d).to(BF).float()).to(BF)
    r = causal_conv3d(y, w, bias, out, ksize, stride, (False, False), res, gn_sums)
    return True, gn_sums is not None and r[1]


def groupnorm_stats(x, G, sums):
    B, C = x.shape[0], x.shape[-1]
    xf = x.double().reshape(B, -1, G, C // G)
    sums[:, :, 0] = xf.sum((1, 3))
    sums[:, :, 1] = (xf * xf).sum((1, 3))
    return sums


def groupnorm_apply(x, sums, gamma, beta, out, G, eps=1e-6, silu=True):
    B, C = x.shape[0], x.shape[-1]
    n = x.numel() // (B * G)
    mean = sums[:, :, 0] / n
    var = (sums[:, :, 1] / n - mean * mean).clamp_min(0)
    rstd = torch.rsqrt(var.float() + eps)
    shp = [B] + [1] * (x.ndim - 2) + [C]
    mean_c = mean.float().repeat_interleave(C // G, 1).reshape(shp)
    rstd_c = rstd.repeat_interleave(C // G, 1).reshape(shp)
    y = ((x.float() - mean_c) * rstd_c * gamma.float() + beta.float()).to(BF).float()
    if silu:
        y = F.silu(y)
    out.copy_(y.to(out.dtype))
    return out


def masked_softmax(scores, probs, Sk, keys_per_frame, scale):
    Sq = scores.shape[0]
    s = scores[:, :Sk].float() * scale
    if keys_per_frame > 0:
        fq = torch.arange(Sq) // keys_per_frame
        fk = torch.arange(Sk) // keys_per_frame
        s = s.masked_fill(fk[None, :] > fq[:, None], float("-inf"))
    probs.zero_()
    probs[:, :Sk] = torch.softmax(s, -1).to(probs.dtype)
    return probs


def blend(a, b, extent, dim):
    """osk_blend_bf16: f32 cross-fade written into b, one rounding"""
    dim = dim % b.ndim
    extent = min(a.shape[dim], b.shape[dim], extent)
    if extent == 0:
        return b
    assert a.is_contiguous() and b.is_contiguous()
    shape = [1] * b.ndim
    shape[dim] = extent
    w = (torch.arange(extent, dtype=torch.float32) / extent).view(shape)
    ia, ib = [slice(None)] * b.ndim, [slice(None)] * b.ndim
    ia[dim] = slice(a.shape[dim] - extent, a.shape[dim])
    ib[dim] = slice(0, extent)
    b[tuple(ib)] = (a[tuple(ia)].float() * (1 - w) + b[tuple(ib)].float() * w).to(b.dtype)
    return b


def attention_hd512_workspace(B, S, device):
    return None


def attention_hd512(q, k, vt, bias_v, out, keys_per_frame, scale, workspace=None):
    """osk_attention_hd512_fwd_bf16: f32 frame-causal softmax(q k^T scale) v (+ bias), P rounded to bf16 before P.V"""
    B, S, C = q.shape
    f = torch.arange(S) // (keys_per_frame if keys_per_frame > 0 else S)
    mask = f[None, :] > f[:, None]
    for b in range(B):
        s_ = (q[b].float() @ k[b].float().T) * scale
        s_ = s_.masked_fill(mask, float("-inf"))
        m = s_.amax(-1, keepdim=True)
        e = torch.exp(s_ - m)
        o = (e.to(torch.bfloat16).float() @ vt[b, :, :S].float().T) / e.sum(-1, keepdim=True)
        if bias_v is not None:
            o = o + bias_v.float()
        out[b].copy_(o.to(out.dtype))
    return out
