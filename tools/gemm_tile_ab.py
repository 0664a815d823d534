#!/usr/bin/env python
"""Same-process A/B of the three bf16 GEMM tile kernels (256 x 256 gemm256x, 256 x 128 gemm256p, 128 x 128 gemm_bf16) at shapes where the
dispatcher's estimate (csrc/gemm_bf16.hip::tile_choice: rounds of the chip x per-tile rate) has to choose: CFG batch 1 and the rows a
sequence-parallel rank holds.  One JSON line per shape: ms per launch of each kernel (osk_gemm_tile_override), what the estimate picks."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_sora_amd import _C

dev = torch.device("cuda")
L = 16896
SHAPES = [(m, n, k, f"{what}, {rows}") for m, rows in ((L, "B=1"), (3 * L // 2, "B=3 P=2 rank"), (3 * L // 4, "B=3 P=4 rank"), (3 * L // 8, "B=3 P=8 rank"), (L // 4, "B=1 P=4 rank"))
          for n, k, what in ((1152, 1152, "proj"), (1152, 4608, "mlp down"), (1152, 5760, "linear2"), (3456, 1152, "qkv"), (4608, 1152, "mlp up"), (8064, 1152, "linear1"))]
for m, n, k, name in SHAPES:
    a = torch.randn(1, m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    b = torch.zeros(n, device=dev)
    out = torch.empty(1, m, n, dtype=torch.bfloat16, device=dev)
    rec = {"shape": [m, n, k], "what": name, "estimate_picks": _C.lib.osk_gemm_tile_choice(m, n, k), "ms": {}}
    for kind, label in ((2, "256x256"), (1, "256x128"), (0, "128x128")):
        _C.lib.osk_gemm_tile_override(kind)
        for _ in range(3):
            _C.gemm(a, w, b, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            _C.gemm(a, w, b, out)
        e1.record()
        torch.cuda.synchronize()
        rec["ms"][label] = round(e0.elapsed_time(e1) / 10, 4)
    _C.lib.osk_gemm_tile_override(-1)
    rec["best"] = min(rec["ms"], key=rec["ms"].get)
    rec["estimate_label"] = {2: "256x256", 1: "256x128", 0: "128x128"}[rec["estimate_picks"]]
    rec["loss_of_estimate"] = round(rec["ms"][rec["estimate_label"]] / rec["ms"][rec["best"]] - 1, 4)
    print(json.dumps(rec), flush=True)
