#!/usr/bin/env python
"""TFLOP/s of osk_gemm_bf16 at MxNxK shapes: median of 7 bursts of 20
launches, random bf16 data.   python tools/gemm_time.py 8192x8192x8192 50688x4608x1152"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.ab_vendor import gemm_arms

for spec in sys.argv[1:]:
    M, N, K = (int(v) for v in spec.split("x"))
    arms, _keep = gemm_arms(M, N, K)
    fn = arms["osk"]
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    res = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        e.synchronize()
        res.append(2.0 * M * N * K * 20 / (s.elapsed_time(e) * 1e-3) / 1e12)
    res.sort()
    print(f"{spec}: {res[len(res) // 2]:.0f} TFLOP/s (min {res[0]:.0f}, max {res[-1]:.0f})", flush=True)
