#!/usr/bin/env python
"""Per-rank kernel times of BASELINE configs[3] (XL, 51 x 720p latent, L = 184,112 tokens, SP = 8) measured on ONE GPU:
the attention launch of one rank (23,014 local query rows x the 184,112 gathered keys in 8 segments) and the block
GEMMs at the rank's row count.  No collectives: this is the compute a rank does per block, not a scaling result."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_sora_amd import _C
from tools.microbench import timeit

BF, DEV = torch.bfloat16, "cuda"
P, Lloc, H, hd, D = 8, 23014, 16, 72, 1152
for B in (1, 3):
    q = torch.randn(B, Lloc, D, device=DEV).to(BF)
    k = torch.randn(P, B, Lloc, D, device=DEV).to(BF)
    v = torch.randn(P, B, Lloc, D, device=DEV).to(BF)
    segp = (Lloc + 63) // 64 * 64
    vts = torch.empty(P, B, H, hd, segp, dtype=BF, device=DEV)
    _C.v_transpose(v.view(P * B, Lloc, D), vts.view(P * B, H, hd, segp), H, hd)
    out = torch.empty(B, Lloc, D, dtype=BF, device=DEV)
    ws = _C.attention_workspace(q.device)
    ms = timeit(lambda: _C.attention_fwd(q, k[0], vts, out, H, hd, hd ** -0.5, n_seg=P, seg_len=Lloc, k_seg_stride=k.stride(0),
                                         vt_seg_stride=vts.stride(0), workspace=ws), iters=3, warm=1)
    fl = 4.0 * B * H * Lloc * (P * Lloc) * hd
    print(json.dumps({"kernel": "attention (one rank of SP=8, 51x720p)", "B": B, "Lq": Lloc, "Lk": P * Lloc, "ms": round(ms, 3),
                      "tflops": round(fl / ms / 1e9, 1)}), flush=True)
    M = B * Lloc
    tot = 0.0
    for (N, K, n_per_step, tag) in [(3 * D, D, 9, "qkv"), (D, D, 9, "proj"), (4 * D, D, 9, "mlp_up"), (D, 4 * D, 9, "mlp_down"),
                                    (7 * D, D, 19, "linear1"), (D, 5 * D, 19, "linear2")]:
        a = torch.randn(1, M, K, device=DEV).to(BF)
        w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(BF)
        o = torch.empty(1, M, N, dtype=BF, device=DEV)
        g = timeit(lambda: _C.gemm(a, w, None, o), iters=10, warm=3)
        tot += g * n_per_step
    print(json.dumps({"B": B, "gemm_ms_per_step_img_stream": round(tot, 2), "attention_ms_per_step": round(ms * 28, 1)}), flush=True)
