"""TEST INFRASTRUCTURE ONLY — CPU restatement (torch, any float dtype) of the reference's Hunyuan causal 3-D VAE.

Not the product and never imported by it: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
use this file, as the checker / CPU baseline.  It is pinned against the REAL reference (imported from
/root/reference by oracle/ref_loader.py) in tests/test_oracle_vs_reference.py and against the committed goldens
(tests/golden/vae_*.npz, made by oracle/make_golden.py from the real reference) in tests/test_oracle_golden.py.

Everything is a pure function of a state dict with the reference's parameter names and NCTHW tensors.
Each function cites the reference lines it follows (paths relative to /root/reference/opensora/models).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

SCALE_FACTOR = 0.476986  # AutoEncoder3DConfig.scale_factor, autoencoder_kl_causal_3d.py:67
SHIFT_FACTOR = 0.0


def causal_conv3d(sd, name, x, stride=(1, 1, 1)):
    """CausalConv3d.forward (hunyuan_vae/unet_causal_3d_blocks.py:63-96): replicate pad (W k//2, H k//2, T k-1 in
    front, 0 behind) then an unpadded conv3d with the given stride.  `name` is the prefix of `.conv.weight`."""
    w, b = sd[name + ".conv.weight"], sd[name + ".conv.bias"]
    k = w.shape[-1]
    if k > 1:
        x = F.pad(x, (k // 2, k // 2, k // 2, k // 2, k - 1, 0), mode="replicate")
    return F.conv3d(x, w, b, stride=stride)


def group_norm(sd, name, x, groups=32, eps=1e-6):
    return F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps)


def resnet_block(sd, name, x, groups=32):
    """ResnetBlockCausal3D.forward (unet_causal_3d_blocks.py:247-259): GN-SiLU-conv1-GN-SiLU-conv2 (+1x1x1 shortcut)."""
    h = F.silu(group_norm(sd, name + ".norm1", x, groups))
    h = causal_conv3d(sd, name + ".conv1", h)
    h = F.silu(group_norm(sd, name + ".norm2", h, groups))
    h = causal_conv3d(sd, name + ".conv2", h)
    if name + ".conv_shortcut.conv.weight" in sd:
        x = causal_conv3d(sd, name + ".conv_shortcut", x)
    return x + h


def upsample_nearest_causal(x, factor):
    """UpsampleCausal3D.forward, interpolation part (unet_causal_3d_blocks.py:135-150): frame 0 is upsampled in
    H, W only; frames 1.. in T, H, W -> T_out = 1 + f_t (T - 1)."""
    ft, fh, fw = factor
    first = x[:, :, :1].repeat_interleave(fh, 3).repeat_interleave(fw, 4)
    if x.shape[2] == 1:
        return first
    rest = x[:, :, 1:].repeat_interleave(ft, 2).repeat_interleave(fh, 3).repeat_interleave(fw, 4)
    return torch.cat((first, rest), 2)


def frame_causal_mask(T, n_hw, dtype, device):
    """prepare_causal_attention_mask (unet_causal_3d_blocks.py:52-60): key frame <= query frame, additive 0/-inf."""
    f = torch.arange(T * n_hw, device=device) // n_hw
    m = torch.zeros(T * n_hw, T * n_hw, dtype=dtype, device=device)
    return m.masked_fill(f[None, :] > f[:, None], float("-inf"))


def mid_attention(sd, name, x, groups=32):
    """UNetMidBlockCausal3D attention branch (unet_causal_3d_blocks.py:345-351) around diffusers Attention
    (one head of dim C, GroupNorm, biased q/k/v/out, residual)."""
    B, C, T, H, W = x.shape
    tok = x.permute(0, 2, 3, 4, 1).reshape(B, T * H * W, C)
    hn = F.group_norm(tok.transpose(1, 2), groups, sd[name + ".group_norm.weight"], sd[name + ".group_norm.bias"], 1e-6).transpose(1, 2)
    q = F.linear(hn, sd[name + ".to_q.weight"], sd[name + ".to_q.bias"])
    k = F.linear(hn, sd[name + ".to_k.weight"], sd[name + ".to_k.bias"])
    v = F.linear(hn, sd[name + ".to_v.weight"], sd[name + ".to_v.bias"])
    mask = frame_causal_mask(T, H * W, torch.float32, x.device)
    s = (q.float() @ k.float().transpose(1, 2)) * (C ** -0.5) + mask
    o = (torch.softmax(s, -1) @ v.float()).to(x.dtype)
    o = F.linear(o, sd[name + ".to_out.0.weight"], sd[name + ".to_out.0.bias"]) + tok
    return o.reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3)


def mid_block(sd, name, x):
    x = resnet_block(sd, name + ".resnets.0", x)
    x = mid_attention(sd, name + ".attentions.0", x)
    return resnet_block(sd, name + ".resnets.1", x)


def block_strides(n_blocks, time_ratio=4, spatial_ratio=8):
    """(s_t, s_h, s_w) per encoder down block / per decoder up block, None = no resampler
    (hunyuan_vae/vae.py:73-94 and :187-210)."""
    ns, nt = int(math.log2(spatial_ratio)), int(math.log2(time_ratio))
    out = []
    for i in range(n_blocks):
        final = i == n_blocks - 1
        sp = i < ns
        tm = (i >= n_blocks - 1 - nt) and not final
        out.append(((2 if tm else 1), (2 if sp else 1), (2 if sp else 1)) if (sp or tm) else None)
    return out


def encoder(sd, cfg, x):
    """EncoderCausal3D.forward (hunyuan_vae/vae.py:128-155)."""
    ch = list(cfg["block_out_channels"])
    lpb = cfg.get("layers_per_block", 2)
    h = causal_conv3d(sd, "encoder.conv_in", x)
    for i, st in enumerate(block_strides(len(ch))):
        for j in range(lpb):
            h = resnet_block(sd, f"encoder.down_blocks.{i}.resnets.{j}", h)
        if st is not None:
            h = causal_conv3d(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", h, st)
    h = mid_block(sd, "encoder.mid_block", h)
    h = F.silu(group_norm(sd, "encoder.conv_norm_out", h))
    return causal_conv3d(sd, "encoder.conv_out", h)


def decoder(sd, cfg, z):
    """DecoderCausal3D.forward (hunyuan_vae/vae.py:246-277)."""
    ch = list(cfg["block_out_channels"])
    lpb = cfg.get("layers_per_block", 2)
    h = causal_conv3d(sd, "decoder.conv_in", z)
    h = mid_block(sd, "decoder.mid_block", h)
    for i, st in enumerate(block_strides(len(ch))):
        for j in range(lpb + 1):
            h = resnet_block(sd, f"decoder.up_blocks.{i}.resnets.{j}", h)
        if st is not None:
            h = upsample_nearest_causal(h, st)
            h = causal_conv3d(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", h)
    h = F.silu(group_norm(sd, "decoder.conv_norm_out", h))
    return causal_conv3d(sd, "decoder.conv_out", h)


def encode_moments(sd, cfg, x):
    """encoder + quant_conv (autoencoder_kl_causal_3d.py:300-306) -> [B, 2*zc, T', h, w] (mean | logvar)."""
    return F.conv3d(encoder(sd, cfg, x), sd["quant_conv.weight"], sd["quant_conv.bias"])


def encode(sd, cfg, x, sample_posterior=False, generator=None):
    """AutoencoderKLCausal3D.encode without tiling (autoencoder_kl_causal_3d.py:269-317) +
    DiagonalGaussianDistribution (vae.py:280-340): mode() = mean; sample() = mean + exp(0.5 clamp(logvar)) * eps."""
    mean, logvar = encode_moments(sd, cfg, x).chunk(2, 1)
    z = mean
    if sample_posterior:
        std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
        z = mean + std * torch.randn(mean.shape, generator=generator, dtype=mean.dtype)
    return SCALE_FACTOR * (z - SHIFT_FACTOR)


def decode(sd, cfg, z):
    """AutoencoderKLCausal3D.decode/_decode without tiling (autoencoder_kl_causal_3d.py:319-358)."""
    z = z / SCALE_FACTOR + SHIFT_FACTOR
    z = F.conv3d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    return decoder(sd, cfg, z)


# ------------------------------------------------------------------------------------------- tiling
def _blend(a, b, extent, dim):
    """blend_v / blend_h / blend_t (autoencoder_kl_causal_3d.py:360-382): linear cross-fade written into b."""
    extent = min(a.shape[dim], b.shape[dim], extent)
    if extent == 0:
        return b
    w = (torch.arange(extent, dtype=torch.float32, device=b.device) / extent).to(b.dtype)
    shape = [1] * b.ndim
    shape[dim] = extent
    w = w.view(shape)
    ia = [slice(None)] * b.ndim
    ib = [slice(None)] * b.ndim
    ia[dim] = slice(a.shape[dim] - extent, a.shape[dim])
    ib[dim] = slice(0, extent)
    b = b.clone()
    b[tuple(ib)] = a[tuple(ia)] * (1 - w) + b[tuple(ib)] * w
    return b


def tile_params(cfg):
    """tile sizes of AutoencoderKLCausal3D.__init__ (autoencoder_kl_causal_3d.py:139-146)."""
    ss, st = cfg.get("sample_size", 256), cfg.get("sample_tsize", 64)
    return dict(sample=ss, latent=int(ss / (2 ** (len(cfg["block_out_channels"]) - 1))), tsample=st,
                tlatent=st // cfg.get("time_compression_ratio", 4), overlap=cfg.get("tile_overlap_factor", 0.25))


def _spatial_tiled(fn, x, tile, stride, blend_extent, row_limit):
    """The common loop of spatial_tiled_encode / spatial_tiled_decode (autoencoder_kl_causal_3d.py:384-489).
    blend_v / blend_h write into tile b in place (:360-374), so a tile blended from above is what its right
    neighbour blends against; `rows` therefore holds the updated tiles."""
    rows = []
    for i in range(0, x.shape[-2], stride):
        rows.append([fn(x[..., i: i + tile, j: j + tile]) for j in range(0, x.shape[-1], stride)])
    out_rows = []
    for i, row in enumerate(rows):
        out_row = []
        for j in range(len(row)):
            t = row[j]
            if i > 0:
                t = _blend(rows[i - 1][j], t, blend_extent, -2)
            if j > 0:
                t = _blend(row[j - 1], t, blend_extent, -1)
            row[j] = t
            out_row.append(t[..., :row_limit, :row_limit])
        out_rows.append(torch.cat(out_row, -1))
    return torch.cat(out_rows, -2)


def _temporal_tiled(fn, x, tile, stride, blend_extent, t_limit):
    """temporal_tiled_encode / temporal_tiled_decode (autoencoder_kl_causal_3d.py:491-552): tiles of tile+1 frames,
    the first output frame of every later tile dropped, cross-fade over blend_extent frames."""
    row = []
    for i in range(0, x.shape[2], stride):
        t = fn(x[:, :, i: i + tile + 1])
        row.append(t[:, :, 1:] if i > 0 else t)
    out = []
    for i in range(len(row)):
        if i > 0:
            row[i] = _blend(row[i - 1], row[i], blend_extent, 2)
            out.append(row[i][:, :, :t_limit])
        else:
            out.append(row[i][:, :, : t_limit + 1])
    return torch.cat(out, 2)


def encode_moments_tiled(sd, cfg, x, spatial=True, temporal=True):
    """AutoencoderKLCausal3D.encode dispatch (autoencoder_kl_causal_3d.py:291-306) up to the moments."""
    tp = tile_params(cfg)
    ov = tp["overlap"]

    def plain(t):
        return encode_moments(sd, cfg, t)

    def sp(t):
        if spatial and (t.shape[-1] > tp["sample"] or t.shape[-2] > tp["sample"]):
            be = int(tp["latent"] * ov)
            return _spatial_tiled(plain, t, tp["sample"], int(tp["sample"] * (1 - ov)), be, tp["latent"] - be)
        return plain(t)

    if temporal and x.shape[2] > tp["tsample"]:
        be = int(tp["tlatent"] * ov)
        return _temporal_tiled(sp, x, tp["tsample"], int(tp["tsample"] * (1 - ov)), be, tp["tlatent"] - be)
    return sp(x)


def encode_tiled(sd, cfg, x, spatial=True, temporal=True):
    mean, _ = encode_moments_tiled(sd, cfg, x, spatial, temporal).chunk(2, 1)
    return SCALE_FACTOR * (mean - SHIFT_FACTOR)


def decode_tiled(sd, cfg, z, spatial=True, temporal=True):
    """AutoencoderKLCausal3D.decode/_decode dispatch (autoencoder_kl_causal_3d.py:319-358)."""
    tp = tile_params(cfg)
    ov = tp["overlap"]
    z = z / SCALE_FACTOR + SHIFT_FACTOR

    def plain(t):
        return decoder(sd, cfg, F.conv3d(t, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"]))

    def sp(t):
        if spatial and (t.shape[-1] > tp["latent"] or t.shape[-2] > tp["latent"]):
            be = int(tp["sample"] * ov)
            return _spatial_tiled(plain, t, tp["latent"], int(tp["latent"] * (1 - ov)), be, tp["sample"] - be)
        return plain(t)

    if temporal and z.shape[2] > tp["tlatent"]:
        be = int(tp["tsample"] * ov)
        return _temporal_tiled(sp, z, tp["tlatent"], int(tp["tlatent"] * (1 - ov)), be, tp["tsample"] - be)
    return sp(z)


def conv_flops(cfg, T, H, W):
    """SURVEY.md §8(d) VAE unit of work: 2*Cin*Cout*k^3*To*Ho*Wo over every conv + 8*S*C^2 + 4*S^2*C for the
    mid-block attention, for one encode of [1,3,T,H,W] and one decode of its latent.  Returns (enc, dec)."""
    ch = list(cfg["block_out_channels"])
    lpb = cfg.get("layers_per_block", 2)
    zc = cfg.get("latent_channels", 16)

    def conv(ci, co, k, t, h, w):
        return 2.0 * ci * co * k ** 3 * t * h * w

    def res(ci, co, t, h, w):
        return conv(ci, co, 3, t, h, w) + conv(co, co, 3, t, h, w) + (conv(ci, co, 1, t, h, w) if ci != co else 0.0)

    def mid(c, t, h, w):
        s = t * h * w
        return 2 * res(c, c, t, h, w) + 8.0 * s * c * c + 4.0 * s * s * c

    t, h, w = T, H, W
    enc = conv(cfg.get("in_channels", 3), ch[0], 3, t, h, w)
    prev = ch[0]
    for i, st in enumerate(block_strides(len(ch))):
        for j in range(lpb):
            enc += res(prev if j == 0 else ch[i], ch[i], t, h, w)
        prev = ch[i]
        if st is not None:
            t, h, w = (t - 1) // st[0] + 1, (h - 1) // st[1] + 1, (w - 1) // st[2] + 1
            enc += conv(ch[i], ch[i], 3, t, h, w)
    enc += mid(ch[-1], t, h, w) + conv(ch[-1], 2 * zc, 3, t, h, w) + conv(2 * zc, 2 * zc, 1, t, h, w)
    dec = conv(zc, zc, 1, t, h, w) + conv(zc, ch[-1], 3, t, h, w) + mid(ch[-1], t, h, w)
    rev = ch[::-1]
    prev = rev[0]
    for i, st in enumerate(block_strides(len(ch))):
        for j in range(lpb + 1):
            dec += res(prev if j == 0 else rev[i], rev[i], t, h, w)
        prev = rev[i]
        if st is not None:
            t, h, w = 1 + st[0] * (t - 1), h * st[1], w * st[2]
            dec += conv(rev[i], rev[i], 3, t, h, w)
    dec += conv(ch[0], cfg.get("out_channels", 3), 3, t, h, w)
    return enc, dec
