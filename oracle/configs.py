"""TEST INFRASTRUCTURE ONLY — named model/shape configurations shared by goldens, tests and bench."""
from __future__ import annotations

from open_sora_amd.configs import MMDIT  # noqa: F401  (one table; the oracle only reads it)

# tiny geometries for goldens (outputs of the real reference are committed under tests/golden/)
_TINY = dict(in_channels=64, vec_in_dim=48, context_in_dim=96, mlp_ratio=4.0, theta=10000, qkv_bias=True,
             guidance_embed=False, cond_embed=True, depth=2, depth_single_blocks=3)
GOLDEN = {
    # name: (cfg, B, T, h, w, L_txt)
    "hd64_eager_fused": (dict(_TINY, hidden_size=128, num_heads=2, axes_dim=[16, 24, 24], fused_qkv=True, use_liger_rope=False), 2, 2, 5, 7, 40),
    "hd64_liger_split": (dict(_TINY, hidden_size=128, num_heads=2, axes_dim=[16, 24, 24], fused_qkv=False, use_liger_rope=True, guidance_embed=True), 1, 3, 4, 6, 24),
    # hd 72 needs hidden % 64 == 0 for the MFMA GEMM K tiling -> 8 heads (hidden 576), fewer blocks
    "hd72_eager_split": (dict(_TINY, hidden_size=576, num_heads=8, axes_dim=[8, 32, 32], fused_qkv=False, use_liger_rope=False, depth=1, depth_single_blocks=2), 2, 2, 6, 6, 32),
    "hd72_liger_fused": (dict(_TINY, hidden_size=576, num_heads=8, axes_dim=[8, 32, 32], fused_qkv=True, use_liger_rope=True, qkv_bias=False, depth=1, depth_single_blocks=2), 1, 2, 7, 9, 16),
    "hd128_liger_split": (dict(_TINY, hidden_size=256, num_heads=2, axes_dim=[16, 56, 56], fused_qkv=False, use_liger_rope=True), 3, 2, 4, 5, 24),
    "hd128_eager_fused": (dict(_TINY, hidden_size=256, num_heads=2, axes_dim=[16, 56, 56], fused_qkv=True, use_liger_rope=False, cond_embed=False), 1, 1, 9, 11, 8),
}
