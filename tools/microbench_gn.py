#!/usr/bin/env python
"""GroupNorm stats / apply at the VAE's tensor shapes (33 x 256 x 256 decode): per-shape time and HBM rate."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_sora_amd import _C
from tools.microbench import timeit

BF = torch.bfloat16
# (T, H, W, C): decoder stages of the 33x256x256 round trip (latent 9x32x32) and the encoder's first stages
SHAPES = [(33, 256, 256, 128), (33, 256, 256, 256), (17, 128, 128, 256), (17, 128, 128, 512), (9, 64, 64, 512), (9, 32, 32, 512)]
for T, H, W, C in SHAPES:
    x = torch.randn(1, T, H, W, C, device="cuda").to(BF)
    sums = torch.empty(1, 32, 2, dtype=torch.float64, device="cuda")
    gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
    out = torch.empty_like(x)
    n = x.numel()
    ms = timeit(lambda: _C.groupnorm_stats(x, 32, sums), iters=20, warm=5)
    ma = timeit(lambda: _C.groupnorm_apply(x, sums, gamma, beta, out, 32, 1e-6, True), iters=20, warm=5)
    print(json.dumps({"shape": [T, H, W, C], "mb": round(2 * n / 1e6, 1), "stats_ms": round(ms, 4), "stats_gbps": round(2 * n / ms / 1e6, 1),
                      "apply_ms": round(ma, 4), "apply_gbps": round(4 * n / ma / 1e6, 1)}), flush=True)
