#!/usr/bin/env python
"""Per-kernel microbenchmarks at the bench shapes (XL cfg 2 and others): TFLOP/s or GB/s per kernel, HIP-event
timed on torch's current stream.  Writes JSON lines to stdout.  Usage: python tools/microbench.py [--quick]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from tools import _altlib

_altlib.install()   # OSK_ALT_LIB: an experiment library instead of the shipped one
from open_sora_amd import _C

DEV = "cuda"
BF = torch.bfloat16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def emit(**kw):
    print(json.dumps(kw), flush=True)


def bench_gemm(M, N, K, tag):
    a = torch.randn(1, M, K, device=DEV).to(BF)
    w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(BF)
    bias = torch.randn(N, device=DEV)
    out = torch.empty(1, M, N, dtype=BF, device=DEV)
    ms = timeit(lambda: _C.gemm(a, w, bias, out), iters=40, warm=10)
    emit(kernel="gemm_bf16", tag=tag, M=M, N=N, K=K, ms=round(ms, 4),
         tflops=round(2.0 * M * N * K / ms / 1e9, 1))


def bench_gemm_fp8(M, N, K, tag):
    a = torch.randn(1, M, K, device=DEV).to(BF)
    w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(BF)
    bias = torch.randn(N, device=DEV)
    out = torch.empty(1, M, N, dtype=BF, device=DEV)
    a8, sa = _C.quantize_rows_fp8(a)
    w8, sw = _C.quantize_rows_fp8(w)
    ms = timeit(lambda: _C.gemm_fp8(a8, sa, w8, sw, bias, out), iters=40, warm=10)
    emit(kernel="gemm_fp8", tag=tag, M=M, N=N, K=K, ms=round(ms, 4), tflops=round(2.0 * M * N * K / ms / 1e9, 1))
    ms = timeit(lambda: _C.quantize_rows_fp8(a, a8, sa), iters=20, warm=5)
    emit(kernel="quantize_rows_fp8", tag=tag, M=M, K=K, ms=round(ms, 4), gbps=round(3.0 * M * K / ms / 1e6, 1))


def bench_attn(B, H, L, hd, tag):
    D = H * hd
    y = torch.randn(B, L, 3 * D, device=DEV).to(BF)
    if os.environ.get("OSK_BENCH_ZERO"):  # DVFS probe: zero operands draw less power -> higher clocks (never a result)
        y.zero_()
    q, k, v = y[:, :, :D], y[:, :, D:2 * D], y[:, :, 2 * D:]
    Lp = (L + 63) // 64 * 64
    vt = torch.empty(B, H, hd, Lp, dtype=BF, device=DEV)
    _C.v_transpose(v, vt, H, hd)
    out = torch.empty(B, L, D, dtype=BF, device=DEV)
    ms = timeit(lambda: _C.attention_fwd(q, k, vt, out, H, hd, hd ** -0.5), iters=5, warm=2)
    emit(kernel="attention_fwd", tag=tag, B=B, H=H, L=L, hd=hd, ms=round(ms, 4),
         tflops=round(4.0 * B * H * L * L * hd / ms / 1e9, 1))
    ws = _C.attention_workspace(q.device)
    ms = timeit(lambda: _C.attention_fwd(q, k, vt, out, H, hd, hd ** -0.5, workspace=ws), iters=5, warm=2)
    emit(kernel="attention_fwd+tailsplit", tag=tag, B=B, H=H, L=L, hd=hd, ms=round(ms, 4),
         tflops=round(4.0 * B * H * L * L * hd / ms / 1e9, 1),
         parts=_C.lib.osk_attention_tail_split_factor(B, H, L, 1, L, hd, ws.numel()))
    # with a score bound (what the model path passes: FAST body when the key count is a multiple of 64)
    D_ = H * hd
    qn = q.float().view(B, L, H, hd).norm(dim=-1).amax().item() * hd ** -0.5 * 1.4426950408889634
    kn = k.float().view(B, L, H, hd).norm(dim=-1).amax().item()
    ms = timeit(lambda: _C.attention_fwd(q, k, vt, out, H, hd, hd ** -0.5, workspace=ws, score_bound=qn * kn), iters=5, warm=2)
    emit(kernel="attention_fwd+tailsplit+bound", tag=tag, B=B, H=H, L=L, hd=hd, ms=round(ms, 4),
         tflops=round(4.0 * B * H * L * L * hd / ms / 1e9, 1), bound=round(qn * kn, 2))
    ms = timeit(lambda: _C.v_transpose(v, vt, H, hd))
    emit(kernel="v_transpose", tag=tag, ms=round(ms, 4), gbps=round(4.0 * B * L * D / ms / 1e6, 1))
    if hd in (72, 128):
        sv = (v.float().abs().view(B, L, H, hd).amax(dim=(1, 3)) / 448.0).contiguous()
        vt8 = torch.empty(B, H, _C.vt8_rows(hd), Lp, dtype=torch.uint8, device=DEV)
        _C.v_transpose_fp8(v, sv, vt8, H, hd)
        ms = timeit(lambda: _C.attention_fwd_pv8(q, k, vt8, sv, out, H, hd, hd ** -0.5, workspace=ws), iters=5, warm=2)
        emit(kernel="attention_fwd_pv8+tailsplit", tag=tag, B=B, H=H, L=L, hd=hd, ms=round(ms, 4),
             tflops=round(4.0 * B * H * L * L * hd / ms / 1e9, 1))
        ms = timeit(lambda: _C.v_transpose_fp8(v, sv, vt8, H, hd))
        emit(kernel="v_transpose_fp8", tag=tag, ms=round(ms, 4), gbps=round(3.0 * B * L * D / ms / 1e6, 1))


def bench_elementwise(B, L, D, H, hd):
    x = torch.randn(B, L, D, device=DEV).to(BF)
    out = torch.empty_like(x)
    mod = torch.randn(B, 2 * D, device=DEV)
    ms = timeit(lambda: _C.ln_modulate(x, mod[:, :D], mod[:, D:], out, mod.stride(0)))
    emit(kernel="ln_modulate", B=B, L=L, D=D, ms=round(ms, 4), gbps=round(4.0 * B * L * D / ms / 1e6, 1))
    y = torch.randn(B, L, 3 * D, device=DEV).to(BF)
    sc = torch.ones(hd, dtype=BF, device=DEV)
    cos = torch.rand(1, L, hd // 2, device=DEV)
    sin = torch.rand(1, L, hd // 2, device=DEV)
    for mode in (0, 1):
        ms = timeit(lambda: _C.qknorm_rope(y[:, :, :D], y[:, :, D:2 * D], sc, sc, sc, sc, 0, cos, sin, 0, H, hd, mode))
        emit(kernel="qknorm_rope", mode=mode, B=B, L=L, D=D, hd=hd, ms=round(ms, 4), gbps=round(8.0 * B * L * D / ms / 1e6, 1))


def main():
    quick = "--quick" in sys.argv
    gemm_only = "--gemm-only" in sys.argv
    attn_only = "--attn-only" in sys.argv
    torch.manual_seed(0)
    L, D, B = 16896, 1152, 3
    M = B * L
    for (N, K, tag) in [] if attn_only else [(3 * D, D, "xl.qkv"), (4 * D, D, "xl.mlp_up"), (D, 4 * D, "xl.mlp_down"), (D, D, "xl.proj"),
                        (7 * D, D, "xl.linear1"), (D, 5 * D, "xl.linear2")]:
        bench_gemm(M, N, K, tag)
    if not quick:
        bench_gemm(8192, 8192, 8192, "square8k")
        bench_gemm(4096, 4096, 4096, "square4k")
        bench_gemm(26484, 3072 * 3, 3072, "11b.qkv")
    if "--fp8" in sys.argv:
        for (N, K, tag) in [(3 * D, D, "xl.qkv"), (4 * D, D, "xl.mlp_up"), (D, 4 * D, "xl.mlp_down"), (D, D, "xl.proj"),
                            (7 * D, D, "xl.linear1"), (D, 5 * D, "xl.linear2")]:
            bench_gemm_fp8(M, N, K, tag)
        bench_gemm_fp8(8192, 8192, 8192, "square8k")
        bench_gemm_fp8(26484, 3072 * 3, 3072, "11b.qkv")
        return
    if gemm_only:
        return
    bench_attn(3, 16, L, 72, "xl.cfg2.b3")
    bench_attn(1, 16, L, 72, "xl.cfg2.b1")
    bench_attn(1, 24, 8828, 128, "11b.256px.b1")
    bench_attn(1, 18, L, 64, "hd64")
    if not attn_only:
        bench_elementwise(B, L, D, 16, 72)


if __name__ == "__main__":
    main()
