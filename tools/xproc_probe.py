#!/usr/bin/env python
"""Does kernel X of one process disturb kernel Y of ANOTHER process (or another stream) on the same GPU?

Follow-up of tools/sp_race_hunt.py, whose per-call checksum chain put the first divergence of the two-rank forward at rank 0's
batched adaLN GEMV (osk_gemv_tasks_bf16: identical inputs, differing outputs) while rank 1 -- the other process on the GPU --
was inside its large-tile GEMMs.  Here a "victim" loops ONE kernel on fixed inputs and compares every result with the first,
while an "aggressor" (second process, or a second stream of the same process) loops another kernel.

    python tools/xproc_probe.py --victim gemv --aggressor gemm256p --iters 20000
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import _altlib

_altlib.install()   # OSK_ALT_LIB=<.so>: probe another build of the library (tools/make_nopk_lib.sh); spawned children re-run this import
import torch
import torch.multiprocessing as mp

BF = torch.bfloat16


def make_kernel(kind, dev="cuda"):
    """returns (launch(), outputs()) for a kernel on fixed inputs"""
    from open_sora_amd import _C
    g = torch.Generator(device=dev).manual_seed(5)
    if kind == "gemv":          # the adaLN GEMV of the hd72 test model: 180 tasks of 64 rows, K = 576, batch 2
        D, n_layers = 576, 30
        ws = [(torch.randn(384, D, device=dev, generator=g) * D ** -0.5).to(BF) for _ in range(n_layers)]
        bs = [torch.randn(384, device=dev, generator=g).to(BF) for _ in range(n_layers)]
        layers, col = [], 0
        for w, b in zip(ws, bs):
            layers.append((w, b, col))
            col += w.shape[0]
        tasks = _C.GemvTasks(layers, dev)
        x = torch.randn(2, D, device=dev, generator=g)
        out = torch.empty(2, col, dtype=torch.float32, device=dev)
        return (lambda: _C.gemv_tasks(x, tasks, out, act_in=1)), (lambda: [out]), (ws, bs, tasks, x)
    if kind in ("gemm256p", "gemm_small", "gemm256x"):
        M, N, K = {"gemm256p": (320, 1152, 576), "gemm_small": (192, 1152, 576), "gemm256x": (4096, 4096, 1152)}[kind]
        a = torch.randn(1, M, K, device=dev, generator=g).to(BF)
        w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(BF)
        bias = torch.randn(N, device=dev, generator=g)
        out = torch.empty(1, M, N, dtype=BF, device=dev)
        return (lambda: _C.gemm(a, w, bias, out)), (lambda: [out]), (a, w, bias)
    if kind == "attn":          # the sequence-parallel call shape: 2 segments of 160 keys, 160 local queries, B 2, H 8, hd 72
        B, H, hd, Lloc, P = 2, 8, 72, 160, 2
        D = H * hd
        q = torch.randn(B, Lloc, D, device=dev, generator=g).to(BF)
        k_all = torch.randn(P, B, Lloc, D, device=dev, generator=g).to(BF)
        v = torch.randn(P * B, Lloc, D, device=dev, generator=g).to(BF)
        Lp = (Lloc + 63) // 64 * 64
        vt_all = torch.zeros(P, B, H, hd, Lp, dtype=BF, device=dev)
        _C.v_transpose(v, vt_all.view(P * B, H, hd, Lp), H, hd)
        out = torch.empty(B, Lloc, D, dtype=BF, device=dev)
        wsp = _C.attention_workspace(torch.device(dev))
        return (lambda: _C.attention_fwd(q, k_all[0], vt_all, out, H, hd, hd ** -0.5, n_seg=P, seg_len=Lloc, k_seg_stride=k_all.stride(0),
                                         vt_seg_stride=vt_all.stride(0), workspace=wsp)), (lambda: [out]), (q, k_all, vt_all)
    if kind == "ln":
        B, L, D = 2, 160, 576
        x = torch.randn(B, L, D, device=dev, generator=g).to(BF)
        mod = torch.randn(B, 2 * D, device=dev, generator=g)
        out = torch.empty_like(x)
        return (lambda: _C.ln_modulate(x, mod[:, :D], mod[:, D:], out, mod.stride(0))), (lambda: [out]), (x, mod)
    if kind == "qknorm":
        B, L, H, hd = 2, 160, 8, 72
        D = H * hd
        y0 = torch.randn(B, L, 3 * D, device=dev, generator=g).to(BF)
        y = y0.clone()
        sc = (1 + 0.1 * torch.randn(hd, device=dev, generator=g)).to(BF)
        cos = torch.rand(1, L, hd // 2, device=dev, generator=g)
        sin = torch.rand(1, L, hd // 2, device=dev, generator=g)

        def launch():
            y.copy_(y0)
            _C.qknorm_rope(y[:, :, :D], y[:, :, D:2 * D], sc, sc, sc, sc, 64, cos, sin, 0, H, hd, 0)
        return launch, (lambda: [y]), (y0, sc, cos, sin)
    if kind == "matmul":
        a = torch.randn(4096, 4096, device=dev, generator=g).to(BF)
        return (lambda: a @ a), (lambda: []), (a,)
    if kind == "copy":
        s = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
        d = torch.empty_like(s)
        return (lambda: d.copy_(s)), (lambda: []), (s, d)
    raise ValueError(kind)


def aggressor_proc(kind, stop):
    torch.cuda.set_device(0)
    launch, _, keep = make_kernel(kind)
    i = 0
    while not stop.is_set():
        for _ in range(32):
            launch()
        i += 1
        if i % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()


def victim(kind, iters, check_every, agg_stream_kind=None):
    torch.cuda.set_device(0)
    launch, outputs, keep = make_kernel(kind)
    side = None
    if agg_stream_kind:
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            alaunch, _, akeep = make_kernel(agg_stream_kind)
    launch()
    torch.cuda.synchronize()
    first = [t.clone() for t in outputs()]
    flags = torch.zeros(iters, dtype=torch.bool, device="cuda")          # per-iteration mismatch flag, kept on the device:
    snaps = [t.clone() for t in first]                                   # no host synchronisation inside the loop
    t0 = time.time()
    for i in range(iters):
        if side is not None and i % 4 == 0:
            with torch.cuda.stream(side):
                for _ in range(4):
                    alaunch()
        for t in outputs():
            t.zero_()
        launch()
        f = torch.zeros((), dtype=torch.bool, device="cuda")
        for a, b in zip(outputs(), first):
            f = f | (a != b).any()
        flags[i] = f
        for a, sn in zip(outputs(), snaps):
            sn.copy_(torch.where(f, a, sn))
        if check_every and (i + 1) % check_every == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    fl = flags.cpu()
    bad = int(fl.sum())
    details = dict(bad_iters=fl.nonzero().flatten().tolist()[:20])
    if bad:
        for a, b in zip(snaps, first):
            ne = (a != b)
            idx = ne.nonzero()
            d = (a.float() - b.float()).abs()
            details.update(last_bad_n=int(ne.sum()), numel=a.numel(), first=idx[0].tolist(), last=idx[-1].tolist(), max_abs=float(d.max()),
                           dim0=sorted(set(idx[:, 0].tolist()))[:8], cols=sorted(set(idx[:, -1].tolist()))[:40])
    return dict(victim=kind, iters=iters, bad=bad, seconds=round(time.time() - t0, 1), details=details)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--victim", default="gemv")
    ap.add_argument("--aggressor", default="gemm256p", help="kernel of the second PROCESS ('none' for none)")
    ap.add_argument("--stream-aggressor", default="", help="kernel of a second STREAM in the victim's process")
    ap.add_argument("--iters", type=int, default=5000)
    ap.add_argument("--check-every", type=int, default=64)
    args = ap.parse_args()
    ctx = mp.get_context("spawn")
    stop = ctx.Event()
    p = None
    if args.aggressor != "none":
        p = ctx.Process(target=aggressor_proc, args=(args.aggressor, stop), daemon=True)
        p.start()
        time.sleep(10)
    r = victim(args.victim, args.iters, args.check_every, args.stream_aggressor or None)
    stop.set()
    if p is not None:
        p.join(timeout=30)
        if p.is_alive():
            p.kill()
    r.update(aggressor=args.aggressor, stream_aggressor=args.stream_aggressor)
    print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
