#!/usr/bin/env python
"""Launch only the attention kernel at the bench shape a few times (target of rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_sora_amd import _C
B, H, L, hd = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (3, 16, 16896, 72)))
D = H * hd
torch.manual_seed(0)
y = torch.randn(B, L, 3 * D, device="cuda").to(torch.bfloat16)
q, k, v = y[:, :, :D], y[:, :, D:2 * D], y[:, :, 2 * D:]
vt = torch.empty(B, H, hd, (L + 63) // 64 * 64, dtype=torch.bfloat16, device="cuda")
_C.v_transpose(v, vt, H, hd)
out = torch.empty(B, L, D, dtype=torch.bfloat16, device="cuda")
pv8 = len(sys.argv) > 5 and sys.argv[5] == "pv8"
if pv8:
    sv = (v.float().abs().view(B, L, H, hd).amax(dim=(1, 3)) / 448.0).contiguous()
    vt8 = torch.empty(B, H, _C.vt8_rows(hd), (L + 63) // 64 * 64, dtype=torch.uint8, device="cuda")
    _C.v_transpose_fp8(v, sv, vt8, H, hd)
# the model path hands the kernel a score bound (QK-norm scales): here the Cauchy-Schwarz bound of the data -> the FAST body runs
qn = q.float().view(B, L, H, hd).norm(dim=-1).amax().item() * hd ** -0.5 * 1.4426950408889634
kn = k.float().view(B, L, H, hd).norm(dim=-1).amax().item()
bound = 0.0 if "nobound" in sys.argv else qn * kn
for _ in range(3):
    if pv8:
        _C.attention_fwd_pv8(q, k, vt8, sv, out, H, hd, hd ** -0.5, workspace=_C.attention_workspace(out.device))
    else:
        _C.attention_fwd(q, k, vt, out, H, hd, hd ** -0.5, workspace=_C.attention_workspace(out.device), score_bound=bound)   # as the model calls it
torch.cuda.synchronize()
