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
