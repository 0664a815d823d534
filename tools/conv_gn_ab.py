#!/usr/bin/env python
"""GN form of the sliding-window conv against the two launches it replaces, layer by layer at the VAE's shapes:
(groupnorm_apply + causal_conv3d) vs (groupnorm_table + causal_conv3d_gn_in), alternating, ms per call.
OSK_ALT_LIB=tools/lib/libosk_gn_<name>.so python tools/conv_gn_ab.py   (variants: tools/make_conv_gn_variants.sh)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import _altlib

lib = _altlib.install()
import torch
from open_sora_amd import _C

SHAPES = [(128, 128, 33, 256, 256), (256, 128, 33, 256, 256), (256, 256, 33, 128, 128), (512, 512, 17, 64, 64)]   # Cin, Cout, T, H, W
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(3)
BF = torch.bfloat16
for ci, co, T, H, W in SHAPES:
    x = torch.randn(1, T, H, W, ci, device=dev, generator=g).to(BF)
    w = (torch.randn(co, 27 * ci, device=dev, generator=g) * (27 * ci) ** -0.5).to(BF)
    b = torch.zeros(co, device=dev)
    gamma, beta = torch.ones(ci, device=dev), torch.zeros(ci, device=dev)
    sums = _C.groupnorm_stats(x, 32, torch.empty(1, 32, 2, dtype=torch.float64, device=dev))
    y, out = torch.empty_like(x), torch.empty(1, T, H, W, co, dtype=BF, device=dev)
    table = torch.empty(1, ci // 8, 16, device=dev)

    def plain():
        _C.groupnorm_apply(x, sums, gamma, beta, y, 32, 1e-6, True)
        _C.causal_conv3d(y, w, b, out, 3)

    def conv_only():
        _C.causal_conv3d(y, w, b, out, 3)

    def fold():
        _C.groupnorm_table(sums, gamma, beta, table, T * H * W, 32, 1e-6)
        assert _C.causal_conv3d_gn_in(x, table, w, b, out, 3)[0]

    res = {}
    for rnd in range(3):
        for name, fn in (("apply+conv", plain), ("conv", conv_only), ("fold", fold)):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 4)
    print(json.dumps({"lib": os.path.basename(lib) if lib else "shipped", "shape": [ci, co, T, H, W],
                      **{k: round(min(v), 4) for k, v in res.items()}}), flush=True)
