#!/usr/bin/env python
"""TIMING EXPERIMENT (wrong numerics for fused epilogues): the bench step with every large Linear routed to
torch.matmul (hipBLASLt) instead of osk_gemm_bf16, to see what the vendor GEMM costs INSIDE the denoise step -- i.e.
whether a faster stand-alone GEMM kernel would show up there at all.  Run under rocprofv3 --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_sora_amd import _C, configs, mmdit, sampling

real_gemm = _C.gemm
_tmp = {}


def vendor_gemm(a, w, bias, out, *, res=None, gate=None, gate_batch_stride=0, gelu_from=None):
    B, L, K = a.shape
    N = w.shape[0]
    if B * L < 256 or N < 128 or a.stride(2) != 1 or w.stride(1) != 1 or out.dtype != torch.bfloat16:
        return real_gemm(a, w, bias, out, res=res, gate=gate, gate_batch_stride=gate_batch_stride, gelu_from=gelu_from)
    key = (B * L, N)
    t = _tmp.get(key)
    if t is None:
        t = _tmp[key] = torch.empty(B * L, N, dtype=torch.bfloat16, device=a.device)
    a2 = a.reshape(B * L, K) if a.is_contiguous() else a.contiguous().view(B * L, K)
    torch.matmul(a2, w.t(), out=t)
    out.copy_(t.view(B, L, N))     # keep live (random-like) data flowing: zero / stale activations would clock every later kernel higher
    return out


if os.environ.get("VENDOR", "1") == "1":
    _C.gemm = vendor_gemm
dev = torch.device("cuda", 0)
cfg = dict(configs.MMDIT["XL"])
torch.manual_seed(1234)
model = mmdit.Flux(device_map=dev, torch_dtype=torch.bfloat16, **cfg)
T, hw, nb, L_txt = 16, 64, 3, 512
L_img = T * (hw // 2) ** 2
g = torch.Generator(device=dev).manual_seed(42)
img = torch.randn(nb, L_img, 64, device=dev, generator=g).to(torch.bfloat16)
txt = (torch.randn(nb, L_txt, 4096, device=dev, generator=g) * 0.2).to(torch.bfloat16)
y_vec = torch.randn(nb, 768, device=dev, generator=g).to(torch.bfloat16)
img_ids, txt_ids = sampling.prepare_ids(nb, T, hw, hw, L_txt, dev, torch.bfloat16)
cond = torch.zeros(nb, L_img, 68, device=dev, dtype=torch.bfloat16)
t_vec = torch.full((nb,), 0.7, dtype=torch.bfloat16, device=dev)
with torch.inference_mode():
    for _ in range(2):
        model(img=img, img_ids=img_ids, txt=txt, txt_ids=txt_ids, timesteps=t_vec, y_vec=y_vec, cond=cond)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        model(img=img, img_ids=img_ids, txt=txt, txt_ids=txt_ids, timesteps=t_vec, y_vec=y_vec, cond=cond)
    torch.cuda.synchronize()
print("forward ms", (time.perf_counter() - t0) / 4 * 1e3, "vendor" if os.environ.get("VENDOR", "1") == "1" else "osk")
