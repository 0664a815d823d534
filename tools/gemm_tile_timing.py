#!/usr/bin/env python
"""Where a gemm256x_kernel tile's time goes, per shape: OSK_ALT_LIB=tools/lib/libosk_gemm_timing.so python tools/gemm_tile_timing.py
(tools/make_gemm_timing_lib.sh).  s_memtime ticks of wave 0 of every workgroup, summed over the launch: address set-up before the
asm statement, the asm statement (cold start + K loop), the epilogue.  One JSON line per shape: ticks per tile (on MI355X s_memtime advances at about the
shader clock -- a v_mul_f32 stream measures 4.5 ticks per instruction, tools/valu_rate_probe.hip -- so ticks read as cycles) and shares."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import _altlib

lib_path = _altlib.install()
assert lib_path, "run with OSK_ALT_LIB=tools/lib/libosk_gemm_timing.so"
import torch
from open_sora_amd import _C

rd = _C.lib.osk_gemm_tile_timing_read          # (the same loaded module that owns the device-side counters)
rd.restype = ctypes.c_int
rd.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
dev = torch.device("cuda")
M = 3 * 16896
# (M, N, K, name, epilogue class): plain = bias only; gelu = GELU over every column (the MLP up-projection); gelu_from = linear1's
# [q|k|v | gelu(mlp)] row (wave-tile-aligned boundary at 3456); gate = gate * x + residual in place (proj, mlp.2, linear2)
SHAPES = [(M, 3456, 1152, "XL qkv", "plain"), (M, 1152, 1152, "XL proj", "gate"), (M, 4608, 1152, "XL mlp up", "gelu"),
          (M, 8064, 1152, "XL linear1", "gelu_from_3456"), (M, 1152, 4608, "XL mlp down", "gate"), (M, 1152, 5760, "XL linear2", "gate"),
          (M, 4608, 1152, "XL mlp up (no GELU)", "plain"), (16896, 8064, 1152, "XL linear1, B=1", "gelu_from_3456"),
          (16896, 1152, 5760, "XL linear2, B=1", "gate"),
          (M, 9216, 3072, "11B qkv", "plain"), (8192, 8192, 8192, "8192^3", "plain")]
only = os.environ.get("OSK_TT_ONLY")          # e.g. OSK_TT_ONLY=gate: only the shapes of that epilogue class
buf = (ctypes.c_ulonglong * 4)()
for m, n, k, name, cls in SHAPES:
    if only and only not in cls:
        continue
    a = torch.randn(1, m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    b = torch.zeros(n, device=dev)
    out = torch.empty(1, m, n, dtype=torch.bfloat16, device=dev)
    kw = {}
    if cls == "gelu":
        kw = dict(gelu_from=0)
    elif cls.startswith("gelu_from_"):
        kw = dict(gelu_from=int(cls.rsplit("_", 1)[1]))
    elif cls == "gate":
        out.normal_()
        kw = dict(res=out, gate=torch.rand(1, n, device=dev), gate_batch_stride=n)
    call = lambda: _C.gemm(a, w, b, out, **kw)
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    rd(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        call()
    e1.record()
    torch.cuda.synchronize()
    rd(buf)
    setup, loop, epi, tiles = (int(x) for x in buf)
    if tiles == 0:      # (the dispatcher chose another tile kernel for this shape: nothing stamped)
        print(json.dumps({"shape": [m, n, k], "what": name, "epilogue": cls, "ms_per_launch": round(e0.elapsed_time(e1) / 4, 4),
                          "note": "not the 256 x 256 tile kernel"}), flush=True)
        continue
    tot = setup + loop + epi
    ms = e0.elapsed_time(e1) / 4
    print(json.dumps({"shape": [m, n, k], "what": name, "epilogue": cls, "ms_per_launch": round(ms, 4), "tflops": round(2 * m * n * k / ms / 1e9, 1),
                      "tiles_per_launch": tiles // 4, "rounds": round(tiles / 4 / 256, 2),
                      "ticks_per_tile": {"setup": round(setup / tiles, 1), "asm_statement": round(loop / tiles, 1), "epilogue": round(epi / tiles, 1)},
                      "share": {"setup": round(setup / tot, 4), "asm_statement": round(loop / tot, 4), "epilogue": round(epi / tot, 4)},
                      "launch_ms_over_rounds_x_tile": round(ms / (-(-tiles // 4 // 256)), 4)}), flush=True)
