"""Model geometries and the algorithmic work formulas used by bench.py (no compute here).

The mounted reference is Open-Sora v2.0: its denoiser is the Flux-style MMDiT.  "S" / "XL"
instantiate that denoiser at DiT-S / DiT-XL width with the reference's 1:2 double:single block ratio; "11B" is
the shipped config (/root/reference/configs/diffusion/inference/256px.py:36-55)."""
from __future__ import annotations

_BASE = dict(in_channels=64, vec_in_dim=768, context_in_dim=4096, mlp_ratio=4.0, theta=10000, qkv_bias=True,
             guidance_embed=False, cond_embed=True, fused_qkv=True, use_liger_rope=False)

MMDIT = {
    "S": dict(_BASE, hidden_size=384, num_heads=6, depth=4, depth_single_blocks=8, axes_dim=[16, 24, 24]),
    "XL": dict(_BASE, hidden_size=1152, num_heads=16, depth=9, depth_single_blocks=19, axes_dim=[8, 32, 32]),
    "11B": dict(_BASE, hidden_size=3072, num_heads=24, depth=19, depth_single_blocks=38, axes_dim=[16, 56, 56],
                fused_qkv=False, use_liger_rope=True),
}


def flops_per_forward(cfg: dict, B: int, L_img: int, L_txt: int) -> float:
    """algorithmic FLOPs of one MMDiT forward (multiply-add = 2; attention = QK^T + PV,
    non-causal, not halved; softmax / norm / activation FLOPs excluded)."""
    D, r = cfg["hidden_size"], cfg["mlp_ratio"]
    nd, ns = cfg["depth"], cfg["depth_single_blocks"]
    L = L_img + L_txt
    C = cfg["in_channels"]
    p2 = cfg.get("patch_size", 2) ** 2
    f = (nd + ns) * ((8 + 4 * r) * B * L * D * D + 4 * B * L * L * D)
    f += 2 * B * L_img * D * (C + (C + p2 if cfg.get("cond_embed") else 0) + C)
    f += 2 * B * L_txt * cfg["context_in_dim"] * D
    f += 2 * B * D * D * (12 * nd + 3 * ns + 2)
    f += 2 * B * (256 + cfg["vec_in_dim"] + 2 * D) * D
    return float(f)


def attention_flops(B: int, H: int, Lq: int, Lk: int, hd: int) -> float:
    return 4.0 * B * H * float(Lq) * float(Lk) * hd


# ---- Hunyuan causal 3-D VAE (the shipped architecture, /root/reference/configs/diffusion/inference/256px.py:57-66)
VAE = {
    "hunyuan": dict(in_channels=3, out_channels=3, latent_channels=16, block_out_channels=(128, 256, 512, 512),
                    layers_per_block=2, norm_num_groups=32, time_compression_ratio=4, spatial_compression_ratio=8),
}


def vae_flops(cfg: dict, T: int, H: int, W: int):
    """SURVEY.md §8(d) VAE unit of work: 2*Cin*Cout*k^3*To*Ho*Wo over every conv plus 8*S*C^2 + 4*S^2*C for the
    mid-block attention, for one encode of [1,3,T,H,W] and one decode of its latent -> (encode, decode) FLOPs."""
    import math

    ch = list(cfg["block_out_channels"])
    lpb, zc = cfg.get("layers_per_block", 2), cfg.get("latent_channels", 16)
    ns, nt = int(math.log2(cfg.get("spatial_compression_ratio", 8))), int(math.log2(cfg.get("time_compression_ratio", 4)))
    n = len(ch)
    strides = []
    for i in range(n):
        sp, tm = i < ns, (i >= n - 1 - nt) and i != n - 1
        strides.append(((2 if tm else 1), (2 if sp else 1), (2 if sp else 1)) if (sp or tm) else None)

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
    for i, st in enumerate(strides):
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
    for i, st in enumerate(strides):
        for j in range(lpb + 1):
            dec += res(prev if j == 0 else rev[i], rev[i], t, h, w)
        prev = rev[i]
        if st is not None:
            t, h, w = 1 + st[0] * (t - 1), h * st[1], w * st[2]
            dec += conv(rev[i], rev[i], 3, t, h, w)
    dec += conv(ch[0], cfg.get("out_channels", 3), 3, t, h, w)
    return enc, dec
