"""MI355X-native causal 3-D VAE (HunyuanVideo VAE) behind the reference's module API.

Mirrors, by name, call signature and state-dict keys (so `hunyuan_vae.safetensors` loads unchanged):
    AutoEncoder3DConfig / AutoencoderKLCausal3D / CausalVAE3D_HUNYUAN
                                   /root/reference/opensora/models/hunyuan_vae/autoencoder_kl_causal_3d.py:59-660
    EncoderCausal3D / DecoderCausal3D / DiagonalGaussianDistribution          hunyuan_vae/vae.py:40-340
    CausalConv3d, ResnetBlockCausal3D, Up/DownsampleCausal3D, UNetMidBlockCausal3D, Up/DownEncoderBlock
                                                                   hunyuan_vae/unet_causal_3d_blocks.py:63-520

The nn.Modules only HOLD parameters.  All arithmetic runs in the gfx950 kernels of include/osk.h through
open_sora_amd/_C.py; there is no eager fallback.  Activations are kept channels-last (NDHWC bf16) between the two
boundary conversions; replicate/causal padding, the nearest upsample and the residual add are folded into the
conv kernel (no padded / upsampled / summed copies are materialised), GroupNorm is one statistics pass + one
fused normalise+SiLU pass, and the mid-block attention evaluates the frame-causal predicate on the fly instead of
building the reference's S x S -inf mask in a Python loop (unet_causal_3d_blocks.py:52-60).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import os
import threading
import weakref

import torch
from torch import Tensor, nn

from . import mmdit as _m  # shares the kernel table (set_ops_for_testing) with the denoiser

BF16 = torch.bfloat16


def _ops():
    return _m.ops()


# =============================================================================================
# parameter containers (names == reference state-dict keys)
# =============================================================================================
class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError(f"{type(self).__name__} holds parameters only; its arithmetic runs in libosk_hip.so "
                           "(AutoencoderKLCausal3D.encode/decode); there is no eager fallback.")


class CausalConv3d(_Holder):
    """unet_causal_3d_blocks.py:63-96 (parameters live at `.conv.weight/.bias`, NCTHW kernel layout)."""

    def __init__(self, chan_in, chan_out, kernel_size, stride=1, **kwargs):
        super().__init__()
        self.kernel_size = kernel_size
        self.stride = (stride,) * 3 if isinstance(stride, int) else tuple(stride)
        self.conv = nn.Conv3d(chan_in, chan_out, kernel_size, stride=self.stride)


class ResnetBlockCausal3D(_Holder):
    def __init__(self, *, in_channels, out_channels=None, groups=32, eps=1e-6, **kwargs):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = CausalConv3d(in_channels, out_channels, 3)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = CausalConv3d(out_channels, out_channels, 3)
        self.conv_shortcut = CausalConv3d(in_channels, out_channels, 1) if in_channels != out_channels else None


class DownsampleCausal3D(_Holder):
    def __init__(self, channels, stride=2):
        super().__init__()
        self.conv = CausalConv3d(channels, channels, 3, stride=stride)


class UpsampleCausal3D(_Holder):
    def __init__(self, channels, upsample_factor=(2, 2, 2)):
        super().__init__()
        self.upsample_factor = tuple(upsample_factor)
        self.conv = CausalConv3d(channels, channels, 3)


class Attention(_Holder):
    """diffusers Attention as configured at unet_causal_3d_blocks.py:312-325 (1 head, GroupNorm, residual)."""

    def __init__(self, dim, groups=32, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, dim, eps=eps, affine=True)
        self.to_q = nn.Linear(dim, dim)
        self.to_k = nn.Linear(dim, dim)
        self.to_v = nn.Linear(dim, dim)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])


class UNetMidBlockCausal3D(_Holder):
    def __init__(self, in_channels, resnet_groups=32, add_attention=True):
        super().__init__()
        self.add_attention = add_attention
        self.attentions = nn.ModuleList([Attention(in_channels, resnet_groups) if add_attention else None])
        self.resnets = nn.ModuleList([ResnetBlockCausal3D(in_channels=in_channels, groups=resnet_groups) for _ in range(2)])


class DownEncoderBlockCausal3D(_Holder):
    def __init__(self, in_channels, out_channels, num_layers, add_downsample, downsample_stride, groups=32):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlockCausal3D(in_channels=in_channels if i == 0 else out_channels, out_channels=out_channels, groups=groups)
            for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([DownsampleCausal3D(out_channels, stride=downsample_stride)]) if add_downsample else None


class UpDecoderBlockCausal3D(_Holder):
    def __init__(self, in_channels, out_channels, num_layers, add_upsample, upsample_scale_factor, groups=32):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlockCausal3D(in_channels=in_channels if i == 0 else out_channels, out_channels=out_channels, groups=groups)
            for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([UpsampleCausal3D(out_channels, upsample_scale_factor)]) if add_upsample else None


def _resample_plan(n_blocks: int, time_ratio: int, spatial_ratio: int):
    """(s_t, s_h, s_w) or None per block (vae.py:73-94, 187-210)."""
    ns, nt = int(math.log2(spatial_ratio)), int(math.log2(time_ratio))
    if time_ratio not in (4, 8):
        raise ValueError(f"Unsupported time_compression_ratio: {time_ratio}.")
    out = []
    for i in range(n_blocks):
        sp = i < ns
        tm = (i < ns) if time_ratio == 8 else ((i >= n_blocks - 1 - nt) and i != n_blocks - 1)
        out.append(((2 if tm else 1), (2 if sp else 1), (2 if sp else 1)) if (sp or tm) else None)
    return out


class EncoderCausal3D(_Holder):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(64,), layers_per_block=2, norm_num_groups=32,
                 act_fn="silu", double_z=True, mid_block_add_attention=True, time_compression_ratio=4,
                 spatial_compression_ratio=8, dropout=0.0):
        super().__init__()
        ch = list(block_out_channels)
        self.conv_in = CausalConv3d(in_channels, ch[0], 3)
        self.strides = _resample_plan(len(ch), time_compression_ratio, spatial_compression_ratio)
        blocks, prev = [], ch[0]
        for i, c in enumerate(ch):
            st = self.strides[i]
            blocks.append(DownEncoderBlockCausal3D(prev, c, layers_per_block, st is not None, st or (1, 1, 1), norm_num_groups))
            prev = c
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = UNetMidBlockCausal3D(ch[-1], norm_num_groups, mid_block_add_attention)
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, ch[-1], eps=1e-6)
        self.conv_out = CausalConv3d(ch[-1], 2 * out_channels if double_z else out_channels, 3)


class DecoderCausal3D(_Holder):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(64,), layers_per_block=2, norm_num_groups=32,
                 act_fn="silu", mid_block_add_attention=True, time_compression_ratio=4, spatial_compression_ratio=8,
                 dropout=0.0):
        super().__init__()
        ch = list(block_out_channels)
        self.conv_in = CausalConv3d(in_channels, ch[-1], 3)
        self.mid_block = UNetMidBlockCausal3D(ch[-1], norm_num_groups, mid_block_add_attention)
        self.factors = _resample_plan(len(ch), time_compression_ratio, spatial_compression_ratio)
        rev = ch[::-1]
        blocks, prev = [], rev[0]
        for i, c in enumerate(rev):
            f = self.factors[i]
            blocks.append(UpDecoderBlockCausal3D(prev, c, layers_per_block + 1, f is not None, f or (1, 1, 1), norm_num_groups))
            prev = c
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, ch[0], eps=1e-6)
        self.conv_out = CausalConv3d(ch[0], out_channels, 3)


class DiagonalGaussianDistribution:
    """hunyuan_vae/vae.py:280-340 (small latent-sized tensors: plain torch on the device)."""

    def __init__(self, parameters: Tensor, deterministic: bool = False):
        if parameters.ndim == 3:
            dim = 2
        elif parameters.ndim in (4, 5):
            dim = 1
        else:
            raise NotImplementedError
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=dim)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, generator=None) -> Tensor:
        noise = torch.randn(self.mean.shape, generator=generator, device=self.parameters.device, dtype=self.parameters.dtype)
        return self.mean + self.std * noise

    def kl(self, other=None) -> Tensor:
        if self.deterministic:
            return torch.Tensor([0.0])
        dims = list(range(1, self.mean.ndim))
        if other is None:
            return 0.5 * torch.sum(torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar, dim=dims)
        return 0.5 * torch.sum(torch.pow(self.mean - other.mean, 2) / other.var + self.var / other.var - 1.0
                               - self.logvar + other.logvar, dim=dims)

    def mode(self) -> Tensor:
        return self.mean


# =============================================================================================
# engine: kernels over NDHWC tensors
# =============================================================================================
def _pad8(c: int) -> int:
    p = 8
    while p < c:
        p *= 2
    return p


class _ConvPlan:
    """weight [Cout, Cin, k, k, k] -> bf16 [Cout, round_up(k^3 * Cin_p, 64)], K = tap-major / channel-minor, Cin
    zero-padded to 8 * 2^j; bias f32."""

    def __init__(self, conv: nn.Conv3d):
        w = conv.weight.detach()
        co, ci, k = w.shape[0], w.shape[1], w.shape[2]
        cip = _pad8(ci)
        wk = torch.zeros(co, k, k, k, cip, dtype=BF16, device=w.device)
        wk[..., :ci] = w.permute(0, 2, 3, 4, 1).to(BF16)
        K = k * k * k * cip
        Kp = (K + 63) // 64 * 64
        self.w = torch.zeros(co, Kp, dtype=BF16, device=w.device)
        self.w[:, :K] = wk.reshape(co, K)
        self.b = None if conv.bias is None else conv.bias.detach().float().contiguous()
        self.cin, self.cin_p, self.cout, self.k = ci, cip, co, k
        self.stride = tuple(conv.stride)


# kernel-side images of the layers' parameters, keyed on the layer object.  A weak map instead of an attribute on the (plain torch)
# nn.Conv3d / nn.GroupNorm / nn.Linear modules: nothing here travels with copy.deepcopy / pickle / torch.save of the VAE (the
# images hold re-laid device weights), and an entry dies with its module.
_PLANS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def _drop_plan(mod, *_):
    """(also the load_state_dict post hook of the layers: a module-level function, so that hooks pickle with the module)"""
    _PLANS.pop(mod, None)


def _plan(mod, kind):
    """kernel-side image of a layer's parameters, cached per parameter-owning layer (nn.Conv3d / nn.GroupNorm / nn.Linear) and
    keyed on the parameters' (storage pointer, in-place version; 0 for inference tensors, which track none): a swapped weight,
    or one rewritten in place through the parameter itself, rebuilds it.  A load_state_dict at ANY level (the layer itself, a
    block, the whole VAE -- also under torch.inference_mode(), where the version key cannot see the copy) drops it through a
    post hook on the layer.  Writes through `p.data` are not visible: call AutoencoderKLCausal3D.invalidate_plan() after them."""
    # the parameter-owning layer: a CausalConv3d (this package's holder or the REFERENCE's own class -- vae_plugin.py runs this
    # engine on the reference's modules) wraps its nn.Conv3d as `.conv`
    leaf = mod.conv if not isinstance(mod, nn.Conv3d) and isinstance(getattr(mod, "conv", None), nn.Conv3d) else mod
    key = tuple((q.data_ptr(), 0 if q.is_inference() else q._version) for q in leaf.parameters())
    c = _PLANS.get(leaf)
    p = c[1] if c is not None and c[0] == key else None
    if p is None:
        if kind == "conv":
            p = _ConvPlan(leaf)
        elif kind == "gn":
            p = (leaf.weight.detach().float().contiguous(), leaf.bias.detach().float().contiguous(), leaf.num_groups, leaf.eps)
        elif kind == "lin":
            p = (leaf.weight.detach().to(BF16).contiguous(), leaf.bias.detach().float().contiguous())
        if _drop_plan not in leaf._load_state_dict_post_hooks.values():
            leaf.register_load_state_dict_post_hook(_drop_plan)
        _PLANS[leaf] = (key, p)
    return p


class _GnSumsPool:
    """The f64 [B, G, 2] accumulators of the fused GroupNorm statistics have to arrive zeroed (the conv epilogue ADDS into them).
    One buffer for a whole encoder / decoder pass, zeroed by ONE fill, handed out slice by slice -- round 3 zeroed one tensor per
    GroupNorm (56 fill launches per encode + decode).  Stream-ordered like every other buffer: the fill precedes the convs."""

    SLOTS = 64

    def __init__(self):
        self.buf, self.used = None, 0

    def begin(self, B: int, G: int, device):
        n = B * G * 2
        self.buf = torch.zeros(self.SLOTS, n, dtype=torch.float64, device=device)
        self.used = 0

    def take(self, B: int, G: int, device) -> Tensor:
        if self.buf is None or self.used >= self.SLOTS or self.buf.shape[1] != B * G * 2 or self.buf.device != torch.device(device):
            return torch.zeros(B, G, 2, dtype=torch.float64, device=device)
        t = self.buf[self.used].view(B, G, 2)
        self.used += 1
        return t

    def end(self):
        self.buf, self.used = None, 0


_GN_POOL = threading.local()


def _gn_pool() -> _GnSumsPool:
    p = getattr(_GN_POOL, "pool", None)
    if p is None:
        p = _GN_POOL.pool = _GnSumsPool()
    return p


def _conv(mod, x: Tensor, up=(False, False), res: Tensor | None = None, gn: int = 0, sums: Tensor | None = None) -> Tensor:
    """gn = G > 0: the output feeds an nn.GroupNorm(G, ...) next -- its statistics are taken in the conv's epilogue
    (osk_causal_conv3d_gn_ndhwc_bf16) and travel with the tensor (`_osk_gn`) to `_gn`, which then skips its read pass.
    sums: a zeroed [B, G, 2] f64 accumulator the caller already took from the pool (nothing launched into it yet)."""
    p = _plan(mod, "conv")
    B, T, H, W, C = x.shape
    assert C == p.cin_p, (C, p.cin_p)
    To, Ho, Wo = _ops().conv_out_dims(T, H, W, p.stride, up)
    out = torch.empty(B, To, Ho, Wo, p.cout, dtype=BF16, device=x.device)
    if gn and p.cout % gn == 0:
        if sums is None:
            sums = _gn_pool().take(B, gn, x.device)
        _, fused = _ops().causal_conv3d(x, p.w, p.b, out, p.k, p.stride, up, res, gn_sums=sums)
        if fused:
            out._osk_gn = (gn, sums)
        return out
    return _ops().causal_conv3d(x, p.w, p.b, out, p.k, p.stride, up, res)


def _gn(mod: nn.GroupNorm, x: Tensor, silu: bool) -> Tensor:
    gamma, beta, G, eps = _plan(mod, "gn")
    have = getattr(x, "_osk_gn", None)
    if have is not None and have[0] == G:
        sums = have[1]
    else:
        sums = torch.empty(x.shape[0], G, 2, dtype=torch.float64, device=x.device)
        _ops().groupnorm_stats(x, G, sums)
    return _ops().groupnorm_apply(x, sums, gamma, beta, torch.empty_like(x), G, eps, silu)


# norm -> SiLU -> conv of the resnet blocks as ONE conv launch that reads the un-normalised tensor (_gn_silu_conv).  Opt-in: on the
# 1.4 kW MI355X the encode + decode takes the same time either way (the transform's VALU work beside the MFMAs costs the convs what
# the apply pass cost the HBM: profiles/r04m_vae_gn_fold_ab.jsonl); what it buys is 16 GB less fabric traffic per encode + decode (90.7 -> 74.2 GB by PMC) and
# one full-resolution activation buffer less.  (OSK_VAE_FOLD_GN=1 in the environment turns it on at import.)
FOLD_GN = bool(os.environ.get("OSK_VAE_FOLD_GN"))


def _gn_silu_conv(norm: nn.GroupNorm, conv, x: Tensor, res: Tensor | None = None, gn: int = 0) -> Tensor:
    """conv(silu(norm(x))) (unet_causal_3d_blocks.py:247-256).  Where the sliding-window kernels take the conv
    (osk_causal_conv3d_gnin_ndhwc_bf16: 3 x 3 x 3, whole 16 x 16 bricks, Cout >= 256 or the two-frame form) the normalised tensor
    never exists: the norm becomes a [B, C / 8, 16] table of per-channel scale / shift pairs and the conv applies it -- same
    rounding points -- while it refills its halo from x.  Otherwise: the apply pass, then the conv."""
    gamma, beta, G, eps = _plan(norm, "gn")
    p = _plan(conv, "conv")
    B, T, H, W, C = x.shape
    have = getattr(x, "_osk_gn", None)
    if have is not None and have[0] == G:
        sums = have[1]
    else:
        sums = torch.empty(B, G, 2, dtype=torch.float64, device=x.device)
        _ops().groupnorm_stats(x, G, sums)
    out_sums = None
    if FOLD_GN and p.k == 3 and tuple(p.stride) == (1, 1, 1) and C == p.cin_p and C % 128 == 0:
        table = torch.empty(B, C // 8, 16, dtype=torch.float32, device=x.device)
        _ops().groupnorm_table(sums, gamma, beta, table, T * H * W, G, eps)
        out = torch.empty(B, T, H, W, p.cout, dtype=BF16, device=x.device)
        out_sums = _gn_pool().take(B, gn, x.device) if gn and p.cout % gn == 0 else None
        ran, fused = _ops().causal_conv3d_gn_in(x, table, p.w, p.b, out, p.k, p.stride, res, gn_sums=out_sums)
        if ran:
            if fused:
                out._osk_gn = (gn, out_sums)
            return out
    # (a conv the folded form declined launched nothing: the pool slot taken for its output statistics is still zero -- hand it on
    # instead of taking a second one, or a decoder pass runs the 64-slot pool dry and falls back to one fill per conv: ADVICE r4)
    h = _ops().groupnorm_apply(x, sums, gamma, beta, torch.empty_like(x), G, eps, True)
    return _conv(conv, h, res=res, gn=gn, sums=out_sums)


def _resnet(blk: ResnetBlockCausal3D, x: Tensor) -> Tensor:
    """ResnetBlockCausal3D.forward (unet_causal_3d_blocks.py:247-259); the residual add rides in conv2's epilogue."""
    h = _gn_silu_conv(blk.norm1, blk.conv1, x, gn=blk.norm2.num_groups)
    sc = x if blk.conv_shortcut is None else _conv(blk.conv_shortcut, x)
    return _gn_silu_conv(blk.norm2, blk.conv2, h, res=sc, gn=blk.norm1.num_groups)   # the next consumer is a GroupNorm of the same grouping


def _mid_attention(att: Attention, x: Tensor) -> Tensor:
    """UNetMidBlockCausal3D attention branch (unet_causal_3d_blocks.py:345-351): GroupNorm -> q,k,v (one head of
    dim C) -> frame-causal softmax(q k^T / sqrt(C)) v -> out proj + residual.  NDHWC == token-major: no rearrange.
    C = 512 (every shipped width): osk_attention_hd512_fwd_bf16 -- scores stay in registers, the mask is a predicate, one
    launch for the whole batch.  Other widths (test geometries): QK^T and P.V as GEMMs around the masked-softmax kernel.
    P V + b_v == P (V + 1 b_v^T) because softmax rows sum to one: the V bias is added after the product either way."""
    o = _ops()
    B, T, H, W, C = x.shape
    S, n_hw = T * H * W, H * W
    Sp = (S + 63) // 64 * 64
    wq, bq = _plan(att.to_q, "lin")
    wk, bk = _plan(att.to_k, "lin")
    wv, bv = _plan(att.to_v, "lin")
    wo, bo = _plan(att.to_out[0], "lin")
    hn = _gn(att.group_norm, x, False).view(B, S, C)
    tok = x.view(B, S, C)
    out = torch.empty_like(tok)
    ones = torch.ones(C, dtype=torch.float32, device=x.device)
    if C == 512:
        q = torch.empty(B, S, C, dtype=BF16, device=x.device)
        k = torch.empty_like(q)
        att_o = torch.empty_like(q)
        vt = torch.zeros(B, C, Sp, dtype=BF16, device=x.device)
        o.gemm(hn, wq, bq, q)
        o.gemm(hn, wk, bk, k)
        for b in range(B):
            o.gemm(wv.view(1, C, C), hn[b], None, vt[b: b + 1, :, :S])     # V^T [C, S] = Wv hn^T
        o.attention_hd512(q, k, vt, bv, att_o, n_hw, C ** -0.5, workspace=o.attention_hd512_workspace(B, S, x.device))
        o.gemm(att_o, wo, bo, out, res=tok, gate=ones, gate_batch_stride=0)
        return out.view(B, T, H, W, C)
    S4 = (S + 3) // 4 * 4
    q = torch.empty(1, S, C, dtype=BF16, device=x.device)
    k = torch.empty_like(q)
    att_o = torch.empty_like(q)
    vt = torch.zeros(1, C, Sp, dtype=BF16, device=x.device)
    scores = torch.empty(1, S, S4, dtype=torch.float32, device=x.device)
    probs = torch.empty(S, Sp, dtype=BF16, device=x.device)
    for b in range(B):
        hb = hn[b: b + 1]
        o.gemm(hb, wq, bq, q)
        o.gemm(hb, wk, bk, k)
        o.gemm(wv.view(1, C, C), hb[0], None, vt[:, :, :S])            # V^T [C, S] = Wv hn^T
        o.gemm(q, k[0], None, scores[:, :, :S])                         # S x S scores, f32
        o.masked_softmax(scores[0], probs, S, n_hw, C ** -0.5)
        o.gemm(probs.view(1, S, Sp), vt[0], bv, att_o)                  # P V (+ b_v)
        o.gemm(att_o, wo, bo, out[b: b + 1], res=tok[b: b + 1], gate=ones, gate_batch_stride=0)
    return out.view(B, T, H, W, C)


def _mid(mid: UNetMidBlockCausal3D, x: Tensor) -> Tensor:
    x = _resnet(mid.resnets[0], x)
    if mid.add_attention:
        x = _mid_attention(mid.attentions[0], x)
    return _resnet(mid.resnets[1], x)


def _to_ndhwc(x: Tensor, c_pad: int) -> Tensor:
    B, C, T, H, W = x.shape
    out = torch.zeros(B, T, H, W, c_pad, dtype=BF16, device=x.device) if c_pad != C else \
        torch.empty(B, T, H, W, C, dtype=BF16, device=x.device)
    out[..., :C].copy_(x.permute(0, 2, 3, 4, 1))
    return out


def _to_ncthw(x: Tensor, dtype) -> Tensor:
    return x.permute(0, 4, 1, 2, 3).contiguous().to(dtype)


def run_encoder(enc: EncoderCausal3D, x: Tensor) -> Tensor:
    """EncoderCausal3D.forward (vae.py:128-155) on NDHWC input (channels padded to 8)."""
    G = enc.conv_norm_out.num_groups
    pool = _gn_pool()
    pool.begin(x.shape[0], G, x.device)
    try:
        h = _conv(enc.conv_in, x, gn=G)
        for blk in enc.down_blocks:
            for r in blk.resnets:
                h = _resnet(r, h)
            if blk.downsamplers is not None:
                h = _conv(blk.downsamplers[0].conv, h, gn=G)
        h = _mid(enc.mid_block, h)
        return _conv(enc.conv_out, _gn(enc.conv_norm_out, h, True))
    finally:
        pool.end()


def run_decoder(dec: DecoderCausal3D, z: Tensor) -> Tensor:
    """DecoderCausal3D.forward (vae.py:246-277); the nearest upsample is folded into the upsampler conv."""
    G = dec.conv_norm_out.num_groups
    pool = _gn_pool()
    pool.begin(z.shape[0], G, z.device)
    try:
        h = _conv(dec.conv_in, z, gn=G)
        h = _mid(dec.mid_block, h)
        for blk in dec.up_blocks:
            for r in blk.resnets:
                h = _resnet(r, h)
            if blk.upsamplers is not None:
                ft, fh, fw = blk.upsamplers[0].upsample_factor
                assert fh == fw and fh in (1, 2) and ft in (1, 2)
                h = _conv(blk.upsamplers[0].conv, h, up=(ft == 2, fh == 2), gn=G)
        return _conv(dec.conv_out, _gn(dec.conv_norm_out, h, True))
    finally:
        pool.end()


# =============================================================================================
# the model
# =============================================================================================
@dataclass
class AutoEncoder3DConfig:
    """Field-for-field autoencoder_kl_causal_3d.py:59-81."""

    from_pretrained: str | None
    act_fn: str = "silu"
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 16
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scale_factor: float = 0.476986
    shift_factor: float = 0
    time_compression_ratio: int = 4
    spatial_compression_ratio: int = 8
    mid_block_add_attention: bool = True
    block_out_channels: tuple = (128, 256, 512, 512)
    sample_size: int = 256
    sample_tsize: int = 64
    use_slicing: bool = False
    use_spatial_tiling: bool = False
    use_temporal_tiling: bool = False
    tile_overlap_factor: float = 0.25
    dropout: float = 0.0
    channel: bool = False


class AutoencoderKLCausal3D(nn.Module):
    """autoencoder_kl_causal_3d.py:84-622: encode / decode / forward / tiling / get_latent_size, same signatures."""

    def __init__(self, config: AutoEncoder3DConfig):
        super().__init__()
        self.config = config
        self.scale_factor, self.shift_factor = config.scale_factor, config.shift_factor
        self.time_compression_ratio = config.time_compression_ratio
        self.spatial_compression_ratio = config.spatial_compression_ratio
        self.z_channels = config.latent_channels
        # (t, h, w) compression of the latent grid; api_fn's non-causal i2v trimming reads compression[0]
        # (/root/reference/opensora/utils/sampling.py:713-722)
        self.compression = (config.time_compression_ratio, config.spatial_compression_ratio,
                            config.spatial_compression_ratio)
        common = dict(block_out_channels=config.block_out_channels, layers_per_block=config.layers_per_block,
                      act_fn=config.act_fn, norm_num_groups=config.norm_num_groups,
                      time_compression_ratio=config.time_compression_ratio,
                      spatial_compression_ratio=config.spatial_compression_ratio,
                      mid_block_add_attention=config.mid_block_add_attention, dropout=config.dropout)
        self.encoder = EncoderCausal3D(in_channels=config.in_channels, out_channels=config.latent_channels, double_z=True, **common)
        self.decoder = DecoderCausal3D(in_channels=config.latent_channels, out_channels=config.out_channels, **common)
        self.quant_conv = nn.Conv3d(2 * config.latent_channels, 2 * config.latent_channels, kernel_size=1)
        self.post_quant_conv = nn.Conv3d(config.latent_channels, config.latent_channels, kernel_size=1)
        self.use_slicing = config.use_slicing
        self.use_spatial_tiling = config.use_spatial_tiling
        self.use_temporal_tiling = config.use_temporal_tiling
        self.tile_sample_min_tsize = config.sample_tsize
        self.tile_latent_min_tsize = config.sample_tsize // config.time_compression_ratio
        self.tile_sample_min_size = config.sample_size
        sample_size = config.sample_size[0] if isinstance(config.sample_size, (list, tuple)) else config.sample_size
        self.tile_latent_min_size = int(sample_size / (2 ** (len(config.block_out_channels) - 1)))
        self.tile_overlap_factor = config.tile_overlap_factor
        self._tile_group = None   # enable_tile_parallel(): process group the tiles of a tiled encode / decode are spread over

    # ---- switches (autoencoder_kl_causal_3d.py:148-190)
    def enable_temporal_tiling(self, use_tiling: bool = True):
        self.use_temporal_tiling = use_tiling

    def disable_temporal_tiling(self):
        self.enable_temporal_tiling(False)

    def enable_spatial_tiling(self, use_tiling: bool = True):
        self.use_spatial_tiling = use_tiling

    def disable_spatial_tiling(self):
        self.enable_spatial_tiling(False)

    def enable_tiling(self, use_tiling: bool = True):
        self.enable_spatial_tiling(use_tiling)
        self.enable_temporal_tiling(use_tiling)

    def disable_tiling(self):
        self.disable_spatial_tiling()
        self.disable_temporal_tiling()

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def invalidate_plan(self):
        """drop every cached kernel-side weight image (re-laid conv weights, f32 biases): they are rebuilt from the
        parameters on the next call"""
        for m in self.modules():
            _drop_plan(m)

    # the cached weight images alias / copy parameter storage: any operation that replaces or rewrites the
    # parameters (load_state_dict, .to(), .cuda(), .half() ...) must drop them
    def _apply(self, fn, *a, **k):
        self.invalidate_plan()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate_plan()
        return r

    # ---- un-tiled cores on NCTHW tensors
    def _moments(self, x: Tensor) -> Tensor:
        """encoder + quant_conv (autoencoder_kl_causal_3d.py:300-305) -> NCTHW moments in x.dtype."""
        h = run_encoder(self.encoder, _to_ndhwc(x, _pad8(x.shape[1])))
        return _to_ncthw(_conv(self.quant_conv, h), x.dtype)

    def _decode_core(self, z: Tensor) -> Tensor:
        """post_quant_conv + decoder (autoencoder_kl_causal_3d.py:331-332)."""
        h = _conv(self.post_quant_conv, _to_ndhwc(z, _pad8(z.shape[1])))
        return _to_ncthw(run_decoder(self.decoder, h), z.dtype)

    # ---- public API
    def encode(self, x: Tensor, sample_posterior: bool = True, return_posterior: bool = False, generator=None):
        assert len(x.shape) == 5, "The input tensor should have 5 dimensions."
        if self.use_temporal_tiling and x.shape[2] > self.tile_sample_min_tsize:
            posterior = self.temporal_tiled_encode(x)
        elif self.use_spatial_tiling and (x.shape[-1] > self.tile_sample_min_size or x.shape[-2] > self.tile_sample_min_size):
            posterior = self.spatial_tiled_encode(x)
        else:
            if self.use_slicing and x.shape[0] > 1:
                moments = torch.cat([self._moments(s) for s in x.split(1)])
            else:
                moments = self._moments(x)
            posterior = DiagonalGaussianDistribution(moments)
        z = posterior.sample(generator=generator) if sample_posterior else posterior.mode()
        z = self.scale_factor * (z - self.shift_factor)
        return (z, posterior) if return_posterior else z

    def _decode(self, z: Tensor) -> Tensor:
        assert len(z.shape) == 5, "The input tensor should have 5 dimensions."
        if self.use_temporal_tiling and z.shape[2] > self.tile_latent_min_tsize:
            return self.temporal_tiled_decode(z)
        if self.use_spatial_tiling and (z.shape[-1] > self.tile_latent_min_size or z.shape[-2] > self.tile_latent_min_size):
            return self.spatial_tiled_decode(z)
        return self._decode_core(z)

    def decode(self, z: Tensor) -> Tensor:
        z = z / self.scale_factor + self.shift_factor
        if self.use_slicing and z.shape[0] > 1:
            return torch.cat([self._decode(s) for s in z.split(1)])
        return self._decode(z)

    def forward(self, sample: Tensor, sample_posterior: bool = True, generator=None):
        """autoencoder_kl_causal_3d.py:554-572 -> (dec, posterior, z)."""
        z, posterior = self.encode(sample, return_posterior=True, sample_posterior=sample_posterior, generator=generator)
        return self.decode(z), posterior, z

    def get_latent_size(self, input_size):
        out = [(input_size[0] - 1) // self.time_compression_ratio + 1]
        for i in range(1, 3):
            out.append((input_size[i] - 1) // self.spatial_compression_ratio + 1)
        return out

    # ---- tiling (autoencoder_kl_causal_3d.py:360-552): the reference's loops; a cross-fade is ONE kernel launch
    # (osk_blend_bf16, f32 math, in place in b) instead of `extent` slice assignments
    @staticmethod
    def _blend(a: Tensor, b: Tensor, extent: int, dim: int) -> Tensor:
        if b.dtype == BF16 and a.dtype == BF16 and a.is_contiguous() and b.is_contiguous():
            return _ops().blend(a, b, extent, dim)
        # other dtypes / strided tiles (a caller's own tensors): round-trip through contiguous bf16 copies, written back into b
        bb = b.to(BF16).contiguous()
        _ops().blend(a.to(BF16).contiguous(), bb, extent, dim)
        b.copy_(bb)
        return b

    def blend_v(self, a, b, blend_extent):
        return self._blend(a, b, blend_extent, -2)

    def blend_h(self, a, b, blend_extent):
        return self._blend(a, b, blend_extent, -1)

    def blend_t(self, a, b, blend_extent):
        return self._blend(a, b, blend_extent, -3)

    # ---- tile parallelism (SURVEY.md section 8(e) "VAE": the tiles of the reference's own tiled encode / decode are
    # independent units): with a process group set, every rank runs fn on the tiles i % P == rank and the results are
    # exchanged by one broadcast per tile from its owner (RCCL over xGMI; each tile's pixels cross a link once per
    # receiver); the blends and the assembly are replicated, so every rank returns the whole tensor.
    def enable_tile_parallel(self, group=None):
        import torch.distributed as dist

        self._tile_group = group if group is not None else dist.group.WORLD
        return self

    def disable_tile_parallel(self):
        self._tile_group = None

    def _map_tiles(self, fn, tiles, out_shape):
        """[fn(t) for t in tiles], the calls spread over the ranks of the tile group (when one is set and there is more
        than one tile).  out_shape(t) -> shape of fn(t) (needed by the ranks that do not compute it)."""
        group = getattr(self, "_tile_group", None)
        if group is None or len(tiles) < 2:
            return [fn(t) for t in tiles]
        import torch.distributed as dist

        P, rank = dist.get_world_size(group), dist.get_rank(group)
        if P == 1:
            return [fn(t) for t in tiles]
        outs = []
        for i, t in enumerate(tiles):
            owner = i % P
            if owner == rank:
                o = fn(t).contiguous()
            else:
                o = torch.empty(out_shape(t), dtype=t.dtype, device=t.device)
            outs.append(o)
        for i, o in enumerate(outs):   # after all local tiles are queued: the broadcasts overlap the remaining compute
            dist.broadcast(o, src=dist.get_global_rank(group, i % P), group=group)
        return outs

    def _moments_shape(self, t):
        B, _, T, H, W = t.shape
        lt, lh, lw = self.get_latent_size((T, H, W))
        return (B, 2 * self.z_channels, lt, lh, lw)

    def _decoded_shape(self, t):
        B, _, T, H, W = t.shape
        return (B, self.config.out_channels, 1 + self.time_compression_ratio * (T - 1) if T > 1 else 1,
                H * self.spatial_compression_ratio, W * self.spatial_compression_ratio)

    def _spatial_tiles(self, fn, x, tile, stride, blend_extent, row_limit, out_shape=None):
        coords = [(i, j) for i in range(0, x.shape[-2], stride) for j in range(0, x.shape[-1], stride)]
        flat = self._map_tiles(fn, [x[..., i: i + tile, j: j + tile] for i, j in coords], out_shape)
        ncol = len(range(0, x.shape[-1], stride))
        rows = [flat[r * ncol: (r + 1) * ncol] for r in range(len(flat) // ncol)]
        out_rows = []
        for i, row in enumerate(rows):
            out_row = []
            for j, t in enumerate(row):
                if i > 0:
                    t = self.blend_v(rows[i - 1][j], t, blend_extent)
                if j > 0:
                    t = self.blend_h(row[j - 1], t, blend_extent)
                out_row.append(t[..., :row_limit, :row_limit])
            out_rows.append(torch.cat(out_row, dim=-1))
        return torch.cat(out_rows, dim=-2)

    def spatial_tiled_encode(self, x: Tensor, return_moments: bool = False):
        ov = self.tile_overlap_factor
        be = int(self.tile_latent_min_size * ov)
        m = self._spatial_tiles(self._moments, x, self.tile_sample_min_size, int(self.tile_sample_min_size * (1 - ov)),
                                be, self.tile_latent_min_size - be, self._moments_shape)
        return m if return_moments else DiagonalGaussianDistribution(m)

    def spatial_tiled_decode(self, z: Tensor) -> Tensor:
        ov = self.tile_overlap_factor
        be = int(self.tile_sample_min_size * ov)
        return self._spatial_tiles(self._decode_core, z, self.tile_latent_min_size,
                                   int(self.tile_latent_min_size * (1 - ov)), be, self.tile_sample_min_size - be,
                                   self._decoded_shape)

    def _temporal_tiles(self, fn, x, tile, stride, blend_extent, t_limit, out_shape=None, spread=False):
        starts = list(range(0, x.shape[2], stride))
        chunks = [x[:, :, i: i + tile + 1] for i in starts]
        # spread = the chunks themselves are the parallel units (no spatial tiling inside fn: that level spreads otherwise)
        outs = self._map_tiles(fn, chunks, out_shape) if spread else [fn(c) for c in chunks]
        row = [t[:, :, 1:].contiguous() if k > 0 else t for k, t in enumerate(outs)]   # contiguous: blended in place
        out = []
        for i, t in enumerate(row):
            if i > 0:
                t = self.blend_t(row[i - 1], t, blend_extent)
                out.append(t[:, :, :t_limit])
            else:
                out.append(t[:, :, : t_limit + 1])
        return torch.cat(out, dim=2)

    def temporal_tiled_encode(self, x: Tensor):
        ov = self.tile_overlap_factor

        def enc(t):
            if self.use_spatial_tiling and (t.shape[-1] > self.tile_sample_min_size or t.shape[-2] > self.tile_sample_min_size):
                return self.spatial_tiled_encode(t, return_moments=True)
            return self._moments(t)

        be = int(self.tile_latent_min_tsize * ov)
        inner_spatial = self.use_spatial_tiling and (x.shape[-1] > self.tile_sample_min_size or x.shape[-2] > self.tile_sample_min_size)
        m = self._temporal_tiles(enc, x, self.tile_sample_min_tsize, int(self.tile_sample_min_tsize * (1 - ov)), be,
                                 self.tile_latent_min_tsize - be, self._moments_shape, spread=not inner_spatial)
        return DiagonalGaussianDistribution(m)

    def temporal_tiled_decode(self, z: Tensor) -> Tensor:
        ov = self.tile_overlap_factor

        def dec(t):
            if self.use_spatial_tiling and (t.shape[-1] > self.tile_latent_min_size or t.shape[-2] > self.tile_latent_min_size):
                return self.spatial_tiled_decode(t)
            return self._decode_core(t)

        be = int(self.tile_sample_min_tsize * ov)
        inner_spatial = self.use_spatial_tiling and (z.shape[-1] > self.tile_latent_min_size or z.shape[-2] > self.tile_latent_min_size)
        return self._temporal_tiles(dec, z, self.tile_latent_min_tsize, int(self.tile_latent_min_tsize * (1 - ov)), be,
                                    self.tile_sample_min_tsize - be, self._decoded_shape, spread=not inner_spatial)


def CausalVAE3D_HUNYUAN(from_pretrained: str = None, device_map="cuda", torch_dtype: torch.dtype = BF16, **kwargs):
    """Factory with the reference signature (autoencoder_kl_causal_3d.py:625-660); `from_pretrained` takes a
    safetensors path."""
    config = AutoEncoder3DConfig(from_pretrained=from_pretrained, **kwargs)
    with torch.device(device_map):
        model = AutoencoderKLCausal3D(config)
    model = model.to(torch_dtype).eval()
    if from_pretrained:
        from safetensors.torch import load_file

        model.load_state_dict(load_file(from_pretrained, device=str(device_map)), strict=True)
    return model
