#!/usr/bin/env python
"""Where a sliding-window conv tile's time goes, per layer shape: OSK_ALT_LIB=tools/lib/libosk_conv_timing.so python
tools/conv_tile_timing.py [gn]  (tools/make_conv_timing_lib.sh).  s_memtime ticks (shader cycles) of wave 0 of every workgroup,
summed over the launch: address set-up, the asm statement (prologue + K loop), the epilogue.  `gn`: the GN form
(causal_conv3d_gn_in) instead of the plain conv.  One JSON line per shape."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import _altlib

lib_path = _altlib.install()
assert lib_path, "run with OSK_ALT_LIB=tools/lib/libosk_conv_timing.so"
import torch
from open_sora_amd import _C

rd = _C.lib.osk_conv_tile_timing_read
rd.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
rd.restype = ctypes.c_int
gn = len(sys.argv) > 1 and sys.argv[1] == "gn"
cls = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("res", "res_stats", "stats") else "plain"   # epilogue class of the plain conv:
# res = + residual add (the resnet blocks' conv2), stats = + the consumer GroupNorm's statistics (osk_causal_conv3d_gn_ndhwc_bf16)
dev = torch.device("cuda")
BF = torch.bfloat16
SHAPES = [(128, 128, 33, 256, 256), (256, 128, 33, 256, 256), (256, 256, 33, 128, 128), (512, 256, 33, 128, 128), (512, 512, 17, 64, 64)]
buf = (ctypes.c_ulonglong * 4)()
g = torch.Generator(device=dev).manual_seed(3)
for ci, co, T, H, W in SHAPES:
    x = torch.randn(1, T, H, W, ci, device=dev, generator=g).to(BF)
    w = (torch.randn(co, 27 * ci, device=dev, generator=g) * (27 * ci) ** -0.5).to(BF)
    b = torch.zeros(co, device=dev)
    out = torch.empty(1, T, H, W, co, dtype=BF, device=dev)
    table = torch.ones(1, ci // 8, 16, device=dev)

    res = torch.randn(1, T, H, W, co, device=dev, generator=g).to(BF) if "res" in cls else None
    sums = torch.zeros(1, 32, 2, dtype=torch.float64, device=dev) if "stats" in cls else None

    def run():
        if gn:
            assert _C.causal_conv3d_gn_in(x, table, w, b, out, 3)[0]
        elif sums is not None:
            _C.causal_conv3d(x, w, b, out, 3, (1, 1, 1), (False, False), res, gn_sums=sums)
        else:
            _C.causal_conv3d(x, w, b, out, 3, (1, 1, 1), (False, False), res)

    run()
    run()
    torch.cuda.synchronize()
    rd(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        run()
    e1.record()
    torch.cuda.synchronize()
    rd(buf)
    setup, loop, epi, tiles = (int(v) for v in buf)
    tot = setup + loop + epi
    ms = e0.elapsed_time(e1) / 4
    print(json.dumps({"shape": [ci, co, T, H, W], "gn_form": gn, "epilogue": cls, "ms_per_launch": round(ms, 4),
                      "tflops": round(2.0 * ci * co * 27 * T * H * W / ms / 1e9, 1), "tiles_per_launch": tiles // 4,
                      "ticks_per_tile": {"setup": round(setup / tiles, 1), "asm_statement": round(loop / tiles, 1), "epilogue": round(epi / tiles, 1)},
                      "share": {"setup": round(setup / tot, 4), "asm_statement": round(loop / tot, 4), "epilogue": round(epi / tot, 4)}}), flush=True)
