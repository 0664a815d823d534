#!/usr/bin/env python
"""Same-process A/B of the projection launches of a block: the round-5 path (QKV pair / linear1 GEMM + osk_v_transpose_bf16) against
round 6's osk_gemm_group_bf16 (V written directly as V^T, one launch per block).  One JSON line per shape: ms of each arm (median of
HIP-event pairs, arms interleaved)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import _altlib

_altlib.install()
import torch
from open_sora_amd import _C

BF, DEV = torch.bfloat16, "cuda"


def med(fns, iters=15, warm=3):
    for _ in range(warm):
        for f in fns:
            f()
    ev = [[] for _ in fns]
    for _ in range(iters):
        for i, f in enumerate(fns):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f()
            e1.record()
            ev[i].append((e0, e1))
    torch.cuda.synchronize()
    return [sorted(a.elapsed_time(b) for a, b in e)[len(e) // 2] for e in ev]


def shape(name, D, H, hd, B, Lt, Li):
    L, R = Lt + Li, 4 * D
    g = torch.Generator(device=DEV).manual_seed(1)
    xm = torch.randn(B, L, D, device=DEV, generator=g).to(BF)
    Lp = (L + 63) // 64 * 64
    vt = torch.zeros(B, H, hd, Lp, dtype=BF, device=DEV)
    # single block
    w1 = (torch.randn(3 * D + R, D, device=DEV, generator=g) * D ** -0.5).to(BF)
    b1 = torch.randn(3 * D + R, device=DEV, generator=g) * 0.1
    y = torch.empty(B, L, 3 * D + R, dtype=BF, device=DEV)

    def single_old():
        _C.gemm(xm, w1, b1, y, gelu_from=3 * D)
        _C.v_transpose(y[:, :, 2 * D: 3 * D], vt, H, hd)

    def single_new():
        assert _C.gemm_group([dict(a=xm, w=w1, bias=b1, out=y, gelu_from=3 * D, skip=(2 * D, D)),
                              dict(x=xm, w=w1[2 * D: 3 * D], bias=b1[2 * D: 3 * D], vt=vt, vt_pos=0, hd=hd)])

    def single_gemm_only():
        _C.gemm(xm, w1, b1, y, gelu_from=3 * D)

    # double block
    wq = [(torch.randn(3 * D, D, device=DEV, generator=g) * D ** -0.5).to(BF) for _ in range(2)]
    bq = [torch.randn(3 * D, device=DEV, generator=g) * 0.1 for _ in range(2)]
    y3 = torch.empty(B, L, 3 * D, dtype=BF, device=DEV)
    rows = (slice(Lt, L), slice(0, Lt))

    def double_old():
        _C.gemm_pair(dict(a=xm[:, rows[0]], w=wq[0], bias=bq[0], out=y3[:, rows[0]]), dict(a=xm[:, rows[1]], w=wq[1], bias=bq[1], out=y3[:, rows[1]]))
        _C.v_transpose(y3[:, :, 2 * D:], vt, H, hd)

    def double_new():
        tasks = []
        for i, pos in ((0, Lt), (1, 0)):
            tasks.append(dict(a=xm[:, rows[i]], w=wq[i][:2 * D], bias=bq[i][:2 * D], out=y3[:, rows[i]]))
            tasks.append(dict(x=xm[:, rows[i]], w=wq[i][2 * D:], bias=bq[i][2 * D:], vt=vt, vt_pos=pos, hd=hd))
        assert _C.gemm_group(tasks)

    def vt_only():
        assert _C.gemm_group([dict(x=xm, w=w1[2 * D: 3 * D], bias=b1[2 * D: 3 * D], vt=vt, vt_pos=0, hd=hd)])

    def v_gemm_only():
        _C.gemm(xm, w1[2 * D: 3 * D], b1[2 * D: 3 * D], y[:, :, 2 * D: 3 * D])

    def vtr_only():
        _C.v_transpose(y[:, :, 2 * D: 3 * D], vt, H, hd)

    t = med([single_old, single_new, double_old, double_new, single_gemm_only, vt_only, v_gemm_only, vtr_only])
    keys = ["single_old", "single_group", "double_old", "double_group", "linear1_gemm_only", "vt_task_only", "v_gemm_only", "v_transpose_only"]
    print(json.dumps({"shape": name, "B": B, "L": L, "D": D, **{k: round(v, 4) for k, v in zip(keys, t)},
                      "single_gain_ms": round(t[0] - t[1], 4), "double_gain_ms": round(t[2] - t[3], 4)}), flush=True)


if __name__ == "__main__":
    shape("XL B=3", 1152, 16, 72, 3, 512, 16384)
    shape("XL B=1", 1152, 16, 72, 1, 512, 16384)
    shape("11B B=3", 3072, 24, 128, 3, 512, 16384)
    shape("11B 256px B=3", 3072, 24, 128, 3, 512, 8316)
