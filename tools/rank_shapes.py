#!/usr/bin/env python
"""What ONE sequence-parallel rank computes per block at BASELINE's long-sequence configurations, measured on ONE GPU with the
loop bodies the model actually selects there: `seqpar.SeqPar.attention` hands the block's score bound down, so
a rank's attention launch is the bounded FAST body (wide 512-row units for head_dim 72 where they fill the chip) -- in fp8 mode the
fp8 P.V body -- over P key segments; the Linear layers of a block run at the rank's row count.  No collectives: this is the compute
of a rank, not a scaling result.

    cfg3       BASELINE configs[3]: XL  (1152, 16 x 72),  51 x 720p latent, L = 184,112 = 8 x 23,014, B = 1
    cfg4       BASELINE configs[4]: 11B (3072, 24 x 128), 64 x 720p latent, L = 230,912 = 8 x 28,864, B = 1, bf16 and fp8
    768px_11b  the reference's own shipped SP = 8 workload (configs/diffusion/inference/768px.py + plugins/sp.py): L = 76,544 = 8 x 9,568, B = 3

`measure(dev)` is bench.py's `rank_shapes` sub-object (never part of `value`); `python tools/rank_shapes.py` prints one JSON line per shape
(tools/microbench_cfg4.py / microbench_cfg5.py are thin wrappers).  Reference call sites: opensora/models/mmdit/distributed.py:413-422
(ring attention of a rank), :473-495 (head exchange), :580-683 (the SP forward)."""
from __future__ import annotations

import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

BF = torch.bfloat16
LOG2E = 1.4426950408889634
PEAK_BF16, PEAK_FP8 = 2500.0, 5000.0

SHAPES = {
    #            model  D     H   hd   P  Lloc    B
    "cfg3":      ("XL", 1152, 16, 72, 8, 23014, 1),
    "cfg4":      ("11B", 3072, 24, 128, 8, 28864, 1),
    "768px_11b": ("11B", 3072, 24, 128, 8, 9568, 3),
}


def _events(fn, iters, warm):
    for _ in range(warm):
        fn()
    evs = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def _unit_rms(t, H, hd):
    s = t.shape
    t = t.view(*s[:-1], H, hd)
    return (t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-6)).view(s)


def attention_rank(dev, H, hd, P, Lloc, B, layout, fp8=False, iters=5, warm=2):
    """one rank's attention launch.  layout "allgather": [B, L/P] query rows x all H heads against P gathered key segments;
    "heads" (head exchange, H % P == 0): P x B query batches of L/P rows sharing B key sets, H / P heads."""
    from open_sora_amd import _C

    g = torch.Generator(device=dev).manual_seed(5)
    if layout == "heads":
        Hl, Bq, Bkv, kvb = H // P, P * B, B, B
    else:
        Hl, Bq, Bkv, kvb = H, B, B, 0
    D = Hl * hd
    q = (_unit_rms(torch.randn(Bq, Lloc, D, device=dev, generator=g), Hl, hd) * (hd ** -0.5 * LOG2E)).to(BF)
    k = _unit_rms(torch.randn(P, Bkv, Lloc, D, device=dev, generator=g), Hl, hd).to(BF)
    v = torch.randn(P, Bkv, Lloc, D, device=dev, generator=g).to(BF)
    segp = (Lloc + 63) // 64 * 64
    out = torch.empty(Bq, Lloc, D, dtype=BF, device=dev)
    ws = _C.attention_workspace(q.device)
    fl = 4.0 * B * H * Lloc * (P * Lloc) * hd
    if fp8:
        sv = (v.float().abs().view(P, Bkv, Lloc, Hl, hd).amax(dim=(0, 2, 4)) / 448.0).contiguous()
        vt8 = torch.zeros(P, Bkv, Hl, _C.vt8_rows(hd), segp, dtype=torch.uint8, device=dev)
        for s_ in range(P):
            _C.v_transpose_fp8(v[s_], sv, vt8[s_], Hl, hd)
        fn = lambda: _C.attention_fwd_pv8(q, k[0], vt8, sv, out, Hl, hd, hd ** -0.5, n_seg=P, seg_len=Lloc, k_seg_stride=k.stride(0),
                                          vt_seg_stride=vt8.stride(0), q_prescaled=True, kv_batches=kvb, workspace=ws)
        body, rows, parts = f"attn_asm{hd}p8_kernel<general>", 256, None
        peak = round(2.0 / (1.0 / PEAK_BF16 + 1.0 / PEAK_FP8), 1)     # QK^T on the bf16 MFMA, P.V on the fp8 MFMA, equal FLOP shares
    else:
        qn = q.float().view(Bq, Lloc, Hl, hd).norm(dim=-1).amax().item()
        kn = k.float().view(P, Bkv, Lloc, Hl, hd).norm(dim=-1).amax().item()
        bound = 1.02 * qn * kn                                        # what mmdit._score_bound derives from unit QK-norm scales, on the data
        vts = torch.zeros(P, Bkv, Hl, hd, segp, dtype=BF, device=dev)
        _C.v_transpose(v.view(P * Bkv, Lloc, D), vts.view(P * Bkv, Hl, hd, segp), Hl, hd)
        fn = lambda: _C.attention_fwd(q, k[0], vts, out, Hl, hd, hd ** -0.5, n_seg=P, seg_len=Lloc, k_seg_stride=k.stride(0),
                                      vt_seg_stride=vts.stride(0), q_prescaled=True, kv_batches=kvb, workspace=ws, score_bound=bound)
        body = _C.attention_body(hd, P, Lloc, bound)
        parts, rows = _C.attention_launch_shape(Bq, Hl, Lloc, P, Lloc, hd, bound, ws.numel())
        peak = PEAK_BF16
    ms = _events(fn, iters, warm)
    assert torch.isfinite(out.float()).all()
    ach = fl / ms / 1e9
    return {"layout": layout, "body": body, "rows_per_unit": rows, "tail_key_parts": parts, "queries": [Bq, Lloc, Hl], "keys": P * Lloc,
            "ms_per_launch": round(ms, 3), "flops_per_launch": fl,
            "roofline": {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4)}}


def block_gemms_rank(dev, D, M, fp8=False, iters=8, warm=3):
    """the two Linear layers of a single-stream block (24 D^2 FLOP per token, the same count as a double block's five) at a
    rank's M rows: linear1 (N = 3 D + 4 D, GELU from column 3 D) and linear2 (K = 5 D, gate * x + residual, in place)"""
    from open_sora_amd import _C

    g = torch.Generator(device=dev).manual_seed(6)
    R = 4 * D
    a1 = torch.randn(1, M, D, device=dev, generator=g).to(BF)
    w1 = (torch.randn(3 * D + R, D, device=dev, generator=g) * D ** -0.5).to(BF)
    b1 = torch.zeros(3 * D + R, device=dev)
    y = torch.empty(1, M, 3 * D + R, dtype=BF, device=dev)
    w2 = (torch.randn(D, D + R, device=dev, generator=g) * (D + R) ** -0.5).to(BF)
    b2 = torch.zeros(D, device=dev)
    x = torch.randn(1, M, D, device=dev, generator=g).to(BF)
    gate = torch.randn(1, D, device=dev, generator=g)
    fl = 2.0 * M * D * (3 * D + R) + 2.0 * M * (D + R) * D
    if fp8:
        a8, sa = _C.quantize_rows_fp8(a1)
        w18, sw1 = _C.quantize_rows_fp8(w1)
        w28, sw2 = _C.quantize_rows_fp8(w2)

        def fn():
            _C.gemm_fp8(a8, sa, w18, sw1, b1, y, gelu_from=3 * D)      # (its activation was quantised by osk_ln_modulate_fp8: not a GEMM cost)
            h8, sh = _C.quantize_rows_fp8(y[:, :, 2 * D:])
            _C.gemm_fp8(h8, sh, w28, sw2, b2, x, res=x, gate=gate, gate_batch_stride=0)
        peak = PEAK_FP8
    else:
        def fn():
            _C.gemm(a1, w1, b1, y, gelu_from=3 * D)
            _C.gemm(y[:, :, 2 * D:], w2, b2, x, res=x, gate=gate, gate_batch_stride=0)
        peak = PEAK_BF16
    ms = _events(fn, iters, warm)
    ach = fl / ms / 1e9
    return {"rows": M, "ms_per_block": round(ms, 3), "flops_per_block": fl,
            "what": "linear1 (+ GELU boundary) + linear2 (gate * x + residual) of one single-stream block" + (" incl. the activation quantisation of linear2" if fp8 else ""),
            "roofline": {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4)}}


def measure(dev, names=("cfg3", "cfg4", "768px_11b"), quick=False) -> dict:
    from open_sora_amd import configs

    out = {"what": "compute of ONE rank of an SP = 8 run per block, on one GPU, with the loop bodies the model selects there; no collectives "
                   "(a per-rank kernel measurement, not a scaling result); timing: median of HIP-event pairs"}
    it = 3 if quick else 5
    for name in names:
        model, D, H, hd, P, Lloc, B = SHAPES[name]
        cfg = configs.MMDIT[model]
        nblk = cfg["depth"] + cfg["depth_single_blocks"]
        rec = {"model": model, "tokens": P * Lloc, "sp": P, "rows_per_rank": Lloc, "cfg_batch": B, "attention": {}, "gemms": {}}
        for layout in ("heads", "allgather"):
            rec["attention"][layout] = attention_rank(dev, H, hd, P, Lloc, B, layout, iters=it)
            torch.cuda.empty_cache()
        rec["attention"]["auto_mode_picks"] = "heads" if H % P == 0 and P >= 4 else "allgather"
        rec["gemms"]["bf16"] = block_gemms_rank(dev, D, B * Lloc)
        if name == "cfg4":
            rec["attention"]["heads_fp8_pv"] = attention_rank(dev, H, hd, P, Lloc, B, "heads", fp8=True, iters=it)
            torch.cuda.empty_cache()
            rec["gemms"]["fp8"] = block_gemms_rank(dev, D, B * Lloc, fp8=True)
        a = rec["attention"][rec["attention"]["auto_mode_picks"]]
        gms = rec["gemms"]["bf16"]["ms_per_block"]
        rec["rank_step_ms_compute_only"] = round(nblk * (a["ms_per_launch"] + gms), 1)
        rec["attention_share"] = round(a["ms_per_launch"] / (a["ms_per_launch"] + gms), 3)
        out[name] = rec
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    names = sys.argv[1:] or ["cfg3", "cfg4", "768px_11b"]
    res = measure(torch.device("cuda", 0), names)
    for k_, v_ in res.items():
        print(json.dumps({k_: v_}), flush=True)
