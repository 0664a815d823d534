#!/usr/bin/env python
"""GEMM time vs K at fixed M, N: the intercept of the linear fit is the per-tile fixed cost (prologue + epilogue +
launch), the slope the steady-state K-step time.  bf16 and fp8 instantiations of gemm256."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_sora_amd import _C
from tools.microbench import timeit

BF = torch.bfloat16
M = 50688
for N in (4608, 1152):
    for fp8 in (False, True):
        pts = []
        for K in (128, 384, 1152, 2304, 4608, 9216):
            a = torch.randn(1, M, K, device="cuda").to(BF)
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
            out = torch.empty(1, M, N, dtype=BF, device="cuda")
            if fp8:
                a8, sa = _C.quantize_rows_fp8(a); w8, sw = _C.quantize_rows_fp8(w)
                ms = timeit(lambda: _C.gemm_fp8(a8, sa, w8, sw, None, out), iters=30, warm=8)
            else:
                ms = timeit(lambda: _C.gemm(a, w, None, out), iters=30, warm=8)
            pts.append((K, ms))
        (k0, t0), (k1, t1) = pts[2], pts[-1]
        slope = (t1 - t0) / (k1 - k0)
        icpt = t0 - slope * k0
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        rounds = -(-tiles // 256)
        print(json.dumps({"N": N, "fp8": fp8, "ms_by_K": pts, "us_per_kstep64_per_round": round(slope * 64 * 1e3 / rounds, 3),
                          "fixed_us_per_round": round(icpt * 1e3 / rounds, 2), "rounds": rounds,
                          "tflops_at_maxK": round(2.0 * M * N * pts[-1][0] / pts[-1][1] / 1e9, 1)}), flush=True)
