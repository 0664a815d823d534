#!/usr/bin/env python
"""Per-rank kernel times of BASELINE configs[4] (11B geometry, fp8 Linears, 64 x 720p latent read as T_lat = 64:
L = 230,912 tokens, SP = 8) measured on ONE GPU: one rank's attention launch (28,864 query rows x 230,912 keys in 8
segments, 24 heads x 128) and its block GEMMs in fp8 at the rank's row count.  No collectives."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_sora_amd import _C
from tools.microbench import timeit

BF, DEV = torch.bfloat16, "cuda"
P, Lloc, H, hd, D = 8, 28864, 24, 128, 3072
B = 1
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
print(json.dumps({"kernel": "attention bf16 (one rank of SP=8, 11B, 64x720p)", "B": B, "Lq": Lloc, "Lk": P * Lloc, "ms": round(ms, 3),
                  "tflops": round(fl / ms / 1e9, 1)}), flush=True)
# fp8 mode: e4m3 V^T with one scale per (batch, head) over all segments, P.V on the fp8 MFMA
sv = (v.float().abs().view(P, B, Lloc, H, hd).amax(dim=(0, 2, 4)) / 448.0).contiguous()
vt8 = torch.empty(P, B, H, _C.vt8_rows(hd), segp, dtype=torch.uint8, device=DEV)
for s_ in range(P):
    _C.v_transpose_fp8(v[s_], sv, vt8[s_], H, hd)
ms8 = timeit(lambda: _C.attention_fwd_pv8(q, k[0], vt8, sv, out, H, hd, hd ** -0.5, n_seg=P, seg_len=Lloc, k_seg_stride=k.stride(0),
                                          vt_seg_stride=vt8.stride(0), workspace=ws), iters=3, warm=1)
print(json.dumps({"kernel": "attention fp8 P.V (same shape)", "ms": round(ms8, 3), "tflops_equiv": round(fl / ms8 / 1e9, 1)}), flush=True)
M = B * Lloc
tot8 = tot16 = 0.0
for (N, K, n_per_step) in [(3 * D, D, 19), (D, D, 19), (4 * D, D, 19), (D, 4 * D, 19), (7 * D, D, 38), (D, 5 * D, 38)]:
    a = torch.randn(1, M, K, device=DEV).to(BF)
    w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(BF)
    o = torch.empty(1, M, N, dtype=BF, device=DEV)
    a8, sa = _C.quantize_rows_fp8(a); w8, sw = _C.quantize_rows_fp8(w)
    tot8 += n_per_step * (timeit(lambda: _C.gemm_fp8(a8, sa, w8, sw, None, o), iters=10, warm=3) +
                          timeit(lambda: _C.quantize_rows_fp8(a, a8, sa), iters=10, warm=3))
    tot16 += n_per_step * timeit(lambda: _C.gemm(a, w, None, o), iters=10, warm=3)
print(json.dumps({"B": B, "gemm_fp8_plus_quant_ms_per_step_img_stream": round(tot8, 1), "gemm_bf16_ms_per_step": round(tot16, 1),
                  "attention_ms_per_step": round(ms * 57, 1), "attention_fp8pv_ms_per_step": round(ms8 * 57, 1)}), flush=True)
