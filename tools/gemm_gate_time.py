#!/usr/bin/env python
"""ms per launch of the gate * x + residual (in place) GEMM at the CFG-batch-1 XL shapes on a forced tile kernel:
[OSK_ALT_LIB=<other library>] python tools/gemm_gate_time.py [kind = 1]   (1 = gemm256p_kernel: the 32 x 32 accumulator layout's epilogue)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import _altlib
_altlib.install()
import torch
from open_sora_amd import _C

kind = int(sys.argv[1]) if len(sys.argv) > 1 else 1
assert _C.lib.osk_gemm_tile_override(kind) == 0
for m, n, k, name in ((16896, 1152, 1152, "proj"), (16896, 1152, 4608, "mlp down"), (16896, 1152, 5760, "linear2"), (50688, 1152, 1152, "proj, B = 3")):
    a = torch.randn(1, m, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
    b = torch.zeros(n, device="cuda")
    out = torch.randn(1, m, n, device="cuda").to(torch.bfloat16)
    gate = torch.rand(1, n, device="cuda")
    call = lambda: _C.gemm(a, w, b, out, res=out, gate=gate, gate_batch_stride=n)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(json.dumps({"kind": kind, "shape": [m, n, k], "what": name, "ms_per_launch": round(ms, 4), "tflops": round(2 * m * n * k / ms / 1e9, 1)}), flush=True)
_C.lib.osk_gemm_tile_override(-1)
