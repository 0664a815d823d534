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

# ---- Hunyuan causal 3-D VAE: tiny geometries for goldens (AutoEncoder3DConfig field names)
_VAE = dict(in_channels=3, out_channels=3, latent_channels=16, norm_num_groups=32, time_compression_ratio=4,
            spatial_compression_ratio=8)
VAE_GOLDEN = {
    # name: (cfg, B, T, H, W)   video [B, 3, T, H, W]; latent [B, 16, (T-1)//4+1, H/8, W/8]
    "c32_lpb1": (dict(_VAE, block_out_channels=(32, 64, 128, 128), layers_per_block=1), 1, 9, 32, 32),
    "c64_lpb2_rect": (dict(_VAE, block_out_channels=(64, 64, 128, 128), layers_per_block=2), 2, 5, 32, 48),
    "c32_single_frame": (dict(_VAE, block_out_channels=(32, 64, 64, 128), layers_per_block=1), 1, 1, 48, 32),
}
# spatial / temporal tiling (autoencoder_kl_causal_3d.py:384-552): tile 32 px / 8 frames on a 48 x 56 x 17 video
VAE_TILED_GOLDEN = {
    "c32_tiled": (dict(_VAE, block_out_channels=(32, 64, 128, 128), layers_per_block=1, sample_size=32,
                       sample_tsize=8, tile_overlap_factor=0.25), 1, 17, 48, 56),
}
