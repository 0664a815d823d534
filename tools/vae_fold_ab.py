#!/usr/bin/env python
"""Same-process A/B of the GroupNorm fold (hunyuan_vae.FOLD_GN): the bench's VAE workload (encode + decode of [1,3,33,256,256]),
alternating fold off / on, wall time per step and the sum of the conv launches (events).  One JSON line per round.
usage: python tools/vae_fold_ab.py [rounds] [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_sora_amd import _C, configs, hunyuan_vae  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = hunyuan_vae.CausalVAE3D_HUNYUAN(device_map=dev, torch_dtype=torch.bfloat16, **dict(configs.VAE["hunyuan"]))
g = torch.Generator(device=dev).manual_seed(42)
x = (torch.randn(1, 3, 33, 256, 256, device=dev, generator=g) * 0.5).clamp(-1, 1).to(torch.bfloat16)


def run(fold):
    hunyuan_vae.FOLD_GN = fold
    with torch.inference_mode():
        out = model.decode(model.encode(x, sample_posterior=False))
        _C.PROFILE_CONV = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model.decode(model.encode(x, sample_posterior=False))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        prof, _C.PROFILE_CONV = _C.PROFILE_CONV, None
    n = len(prof) // steps
    per = [(sum(prof[i + k * n][0].elapsed_time(prof[i + k * n][1]) for k in range(steps)) / steps, prof[i][2]) for i in range(n)]
    return ms, sum(s.elapsed_time(e) for s, e, _ in prof) / steps, out, per


ref = None
for r in range(rounds):
    rec = {"round": r}
    for fold in (False, True):
        ms, conv_ms, out, per = run(fold)
        rec.setdefault("per", []).append(per)
        rec["fold" if fold else "plain"] = {"ms_per_step": round(ms, 3), "conv_ms": round(conv_ms, 3)}
        if ref is None:
            ref = out.float()
        else:      # (the plain path against itself too: the fused statistics add in LDS-atomic order, so runs differ in the last f32 bits)
            rec["rel_l2_%s_vs_first_plain" % ("fold" if fold else "plain")] = float((out.float() - ref).norm() / ref.norm())
    per = rec.pop("per")
    print(json.dumps(rec), flush=True)
    if r == rounds - 1 and os.environ.get("AB_LAYERS"):      # per conv launch (same order in both modes): ms plain, ms fold, GFLOP
        for i, ((a, f), (b, _)) in enumerate(zip(*per)):
            print(json.dumps({"launch": i, "gflop": round(f / 1e9, 1), "plain_ms": round(a, 4), "fold_ms": round(b, 4), "d_ms": round(b - a, 4)}))
