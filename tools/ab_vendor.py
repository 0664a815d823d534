#!/usr/bin/env python
"""Same-box, same-process, same-shape, random-data A/B of this library's MFMA kernels against the vendor kernels
that ship with PyTorch-ROCm ("prove or retract the power-wall ceiling").

  arm "osk"    : osk_gemm_bf16 (gemm256_kernel) / osk_attention_fwd_ws_bf16 (attn_asm72_kernel)
  arm "vendor" : torch.matmul (hipBLASLt / rocBLAS) / F.scaled_dot_product_attention (the flash / CK backend)

The two arms are INTERLEAVED (A B A B ..., each sample a short burst timed with HIP events on the same stream) so both
see the same thermal / DVFS state; data is random bf16 (zero data clocks higher: a probe, never a result).
Shapes: the five Linear shapes of one XL block at the bench size (M = 3 x 16,896) + 8192^3; attention at the bench
shape (B 3, H 16, L 16,896, hd 72) -- SDPA gets the same q/k/v as [B, H, L, hd] tensors.

  python tools/ab_vendor.py                     -> timing table + JSON (gpurun_out/ab_vendor.json)
  python tools/ab_vendor.py --one gemm:osk:50688x3456x1152   (one arm, a few launches: the target of a
                                                              `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE` pass;
                                                              tools/gpu_ab_vendor.sh drives those and merges the clocks)
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from open_sora_amd import _C

DEV, BF = "cuda", torch.bfloat16
GEMM_SHAPES = [  # (M, N, K, what)
    (50688, 3456, 1152, "double-block QKV (img+txt rows)"),
    (50688, 1152, 1152, "attention out-projection"),
    (50688, 4608, 1152, "MLP up"),
    (50688, 1152, 4608, "MLP down"),
    (50688, 8064, 1152, "single-block linear1 [q|k|v|mlp]"),
    (50688, 1152, 5760, "single-block linear2"),
    (8192, 8192, 8192, "8192^3"),
]
ATTN_SHAPE = (3, 16, 16896, 72)


def gemm_arms(M, N, K):
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = torch.randn(1, M, K, device=DEV, generator=g).to(BF)
    w = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).to(BF)
    out = torch.empty(1, M, N, dtype=BF, device=DEV)
    out2 = torch.empty(M, N, dtype=BF, device=DEV)
    a2, wt = a[0], w.t()
    return {"osk": lambda: _C.gemm(a, w, None, out), "vendor": lambda: torch.matmul(a2, wt, out=out2)}, (out, out2)


def attn_arms(B, H, L, hd):
    D = H * hd
    g = torch.Generator(device=DEV).manual_seed(7)
    y = torch.randn(B, L, 3 * D, device=DEV, generator=g).to(BF)
    q, k, v = y[:, :, :D], y[:, :, D:2 * D], y[:, :, 2 * D:]
    vt = torch.zeros(B, H, hd, (L + 63) // 64 * 64, dtype=BF, device=DEV)
    _C.v_transpose(v, vt, H, hd)
    out = torch.empty(B, L, D, dtype=BF, device=DEV)
    ws = _C.attention_workspace(torch.device(DEV))
    q4, k4, v4 = (t.reshape(B, L, H, hd).transpose(1, 2).contiguous() for t in (q, k, v))
    return {"osk": lambda: _C.attention_fwd(q, k, vt, out, H, hd, hd ** -0.5, workspace=ws),
            "vendor": lambda: F.scaled_dot_product_attention(q4, k4, v4)}, None


def burst(fn, n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    return s, e


def ab(arms, flops, n_samples=8, burst_len=6, warm=4):
    for fn in arms.values():
        for _ in range(warm):
            fn()
    torch.cuda.synchronize()
    ev = {k: [] for k in arms}
    for _ in range(n_samples):
        for k, fn in arms.items():       # A B A B ...
            ev[k].append(burst(fn, burst_len))
    torch.cuda.synchronize()
    res = {}
    for k, lst in ev.items():
        ms = sorted(s.elapsed_time(e) / burst_len for s, e in lst)
        med = ms[len(ms) // 2]
        res[k] = {"ms": round(med, 4), "ms_min": round(ms[0], 4), "ms_max": round(ms[-1], 4),
                  "tflops": round(flops / med / 1e9, 1)}
    return res


def one(spec):
    kind, arm, shape = spec.split(":")
    dims = [int(x) for x in shape.split("x")]
    arms, _ = gemm_arms(*dims) if kind == "gemm" else attn_arms(*dims)
    for _ in range(6):
        arms[arm]()
    torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--one", default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ab_vendor.json"))
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    if args.one:
        return one(args.one)
    rows = []
    with torch.inference_mode():
        for M, N, K, what in (GEMM_SHAPES[:2] if args.quick else GEMM_SHAPES):
            arms, outs = gemm_arms(M, N, K)
            r = ab(arms, 2.0 * M * N * K)
            arms["osk"](); arms["vendor"]()
            torch.cuda.synchronize()
            diff = float((outs[0][0].float() - outs[1].float()).abs().max())
            rows.append({"op": "gemm", "shape": [M, N, K], "what": what, **r, "max_abs_diff_between_arms": round(diff, 4),
                         "vendor_over_osk": round(r["vendor"]["tflops"] / r["osk"]["tflops"], 3)})
            print(json.dumps(rows[-1]), flush=True)
            del arms, outs
        B, H, L, hd = ATTN_SHAPE
        arms, _ = attn_arms(B, H, L, hd)
        r = ab(arms, 4.0 * B * H * L * L * hd, n_samples=6, burst_len=3, warm=2)
        rows.append({"op": "attention", "shape": [B, H, L, hd], "what": "bench attention launch (non-causal, bf16)", **r,
                     "vendor_over_osk": round(r["vendor"]["tflops"] / r["osk"]["tflops"], 3)})
        print(json.dumps(rows[-1]), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump({"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "data": "random bf16",
               "method": "interleaved bursts, HIP events, median of samples", "rows": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
