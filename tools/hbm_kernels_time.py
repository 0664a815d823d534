#!/usr/bin/env python
"""Stand-alone timings of the HBM-bound kernels of the denoise step at the XL bench shape (B = 3, L = 16,896, 16 x 72) and the 11B
shape (24 x 128): QK-norm + RoPE, LN + modulate, V transpose.  One JSON line per kernel: us per launch, algorithmic GB/s."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import _altlib
_altlib.install()
import torch
from open_sora_amd import _C

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B, L = 3, 16896
for H, hd, R in ((16, 72, 4), (24, 128, 4)):
    D = H * hd
    y = torch.randn(B, L, 3 * D + R * D, device="cuda").to(torch.bfloat16)
    q, k, v = y[:, :, :D], y[:, :, D:2 * D], y[:, :, 2 * D:3 * D]
    sc = [torch.ones(hd, device="cuda", dtype=torch.bfloat16) for _ in range(4)]
    cos = torch.rand(B, L, hd // 2, device="cuda"); sin = torch.rand(B, L, hd // 2, device="cuda")
    for mode in (0, 1):
        us = timeit(lambda: _C.lib.osk_qknorm_rope_bf16(q.data_ptr(), k.data_ptr(), q.stride(0), q.stride(1), sc[0].data_ptr(), sc[1].data_ptr(),
                                                        sc[2].data_ptr(), sc[3].data_ptr(), 512, cos.data_ptr(), sin.data_ptr(), cos.stride(0), B, L, H, hd, mode,
                                                        1e-6, 1.0, _C._stream()))
        byts = 2 * 2 * B * L * D * 2
        print(json.dumps(dict(kernel=f"qknorm_rope hd{hd} mode{mode}", us=round(us, 1), gbps=round(byts / us / 1e3, 0))))
    vt = torch.empty(B, H, hd, L, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: _C.v_transpose(v, vt, H, hd))
    print(json.dumps(dict(kernel=f"v_transpose hd{hd}", us=round(us, 1), gbps=round(2 * B * L * D * 2 / us / 1e3, 0))))

# the adaLN modulation of one XL step: all 165 chunks of 1152 rows (9 double blocks x 2 streams x 6, 19 single blocks x 3) as ONE task list
D, nl = 1152, 9 * 2 * 6 + 19 * 3
ws = [torch.randn(D, D, device="cuda").mul(D ** -0.5).to(torch.bfloat16) for _ in range(nl)]
bs_ = [torch.zeros(D, device="cuda", dtype=torch.bfloat16) for _ in range(nl)]
tasks = _C.GemvTasks([(w, b, i * D) for i, (w, b) in enumerate(zip(ws, bs_))], "cuda")
vec = torch.randn(3, D, device="cuda")
mod = torch.empty(3, nl * D, device="cuda")
us = timeit(lambda: _C.gemv_tasks(vec, tasks, mod, act_in=1))
print(json.dumps(dict(kernel=f"gemv_tasks adaLN of an XL step ({nl} x {D} rows, K = {D}, 3 vectors)", us=round(us, 1), gbps=round(nl * D * D * 2 / us / 1e3, 0))))

# LayerNorm + modulate at the XL and 11B widths (whole joint sequence, CFG batch 3)
for Dm in (1152, 3072):
    xx = torch.randn(B, L, Dm, device="cuda").to(torch.bfloat16)
    oo = torch.empty_like(xx)
    mv = torch.randn(B, 2 * Dm, device="cuda") * 0.1
    us = timeit(lambda: _C.ln_modulate(xx, mv[:, :Dm], mv[:, Dm:], oo, mv.stride(0)))
    print(json.dumps(dict(kernel=f"ln_modulate D{Dm}", us=round(us, 1), gbps=round(2 * B * L * Dm * 2 / us / 1e3, 0))))
