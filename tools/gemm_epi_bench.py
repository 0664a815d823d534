#!/usr/bin/env python
"""Time osk_gemm_bf16 at the XL block shapes WITH their real epilogues (bias, GELU columns, gate * x + residual):
the un-fused A/B rows of tools/ab_vendor.py do not see epilogue cost.  JSON lines on stdout."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_sora_amd import _C

DEV, BF = "cuda", torch.bfloat16
M, D, R = 3 * 16896, 1152, 4608
CASES = [("qkv bias", 3 * D, D, None, False), ("proj gate+res", D, D, None, True), ("mlp-up bias+gelu", R, D, 0, False),
         ("mlp-down gate+res", D, R, None, True), ("linear1 bias+gelu(mlp cols)", 3 * D + R, D, 3 * D, False),
         ("linear2 gate+res", D, D + R, None, True)]


def timeit(fn, iters=30, warm=8):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


tot = 0.0
for name, N, K, gelu_from, gated in CASES:
    g = torch.Generator(device=DEV).manual_seed(N + K)
    a = torch.randn(3, M // 3, K, device=DEV, generator=g).to(BF)
    w = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).to(BF)
    bias = torch.randn(N, device=DEV, generator=g) * 0.1
    out = torch.randn(3, M // 3, N, device=DEV, generator=g).to(BF)
    kw = {}
    if gated:
        gate = torch.randn(3, N, device=DEV, generator=g) * 0.5
        kw = dict(res=out, gate=gate, gate_batch_stride=gate.stride(0))
    ms = timeit(lambda: _C.gemm(a, w, bias, out, gelu_from=gelu_from, **kw))
    tot += ms
    print(json.dumps({"case": name, "M": M, "N": N, "K": K, "ms": round(ms, 4), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}), flush=True)
print(json.dumps({"case": "sum of the six", "ms": round(tot, 4)}))
