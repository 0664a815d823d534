#!/usr/bin/env python
"""Launch only the large-tile GEMM at one shape a few times (target of rocprofv3 --pmc passes).
usage: gemm_only.py M N K [fp8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_sora_amd import _C
M, N, K = (int(x) for x in sys.argv[1:4])
fp8 = len(sys.argv) > 4 and sys.argv[4] == "fp8"
torch.manual_seed(0)
a = torch.randn(1, M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
out = torch.empty(1, M, N, dtype=torch.bfloat16, device="cuda")
if fp8:
    a8, sa = _C.quantize_rows_fp8(a); w8, sw = _C.quantize_rows_fp8(w)
for _ in range(5):
    if fp8:
        _C.gemm_fp8(a8, sa, w8, sw, None, out)
    else:
        _C.gemm(a, w, None, out)
torch.cuda.synchronize()
