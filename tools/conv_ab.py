#!/usr/bin/env python
"""Conv layer timings at the VAE's stride-1 shapes: OSK_ALT_LIB=tools/lib/libosk_conv_nosw.so python tools/conv_ab.py  (A/B of
the sliding-window kernel against the implicit-GEMM kernel; tools/make_conv_nosw_lib.sh builds the alternative library)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import _altlib

lib = _altlib.install()
import torch
from open_sora_amd import _C

SHAPES = [  # Cin, Cout, T, H, W (source dims), up
    (128, 128, 33, 256, 256, False), (256, 128, 33, 256, 256, False), (256, 256, 33, 128, 128, False), (512, 256, 33, 128, 128, False),
    (512, 512, 17, 64, 64, False), (512, 512, 9, 32, 32, False),
    (512, 512, 17, 64, 64, True), (256, 256, 33, 128, 128, "hw"),       # the decoder's two big upsample convs (up1: T, H, W; up2: H, W)
]
if os.environ.get("CONV_AB_SHAPES"):   # e.g. "0,3": a subset (PMC passes)
    SHAPES = [SHAPES[int(i)] for i in os.environ["CONV_AB_SHAPES"].split(",")]
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(3)
for ci, co, T, H, W, upm in SHAPES:
    up = (upm is True, bool(upm))
    To, Ho, Wo = _C.conv_out_dims(T, H, W, (1, 1, 1), up)
    x = torch.randn(1, T, H, W, ci, device=dev, generator=g).to(torch.bfloat16)
    K = 27 * ci
    w = (torch.randn(co, (K + 63) // 64 * 64, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.zeros(co, device=dev)
    out = torch.empty(1, To, Ho, Wo, co, dtype=torch.bfloat16, device=dev)
    for _ in range(2):
        _C.causal_conv3d(x, w, b, out, 3, (1, 1, 1), up, None)
    torch.cuda.synchronize()
    n = 6
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        _C.causal_conv3d(x, w, b, out, 3, (1, 1, 1), up, None)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * 27 * ci * co * To * Ho * Wo
    print(json.dumps(dict(lib=os.path.basename(lib) if lib else "shipped", cin=ci, cout=co, T=T, H=H, W=W, up=str(upm), ms=round(ms, 4),
                          tflops=round(fl / ms / 1e9, 1))), flush=True)
