#!/usr/bin/env python
"""Matrix of (victim, aggressor) pairs on two streams of one process: which CU resource lets one kernel disturb another?
Victims / synthetic aggressors: tools/interfere_probe.hip (built into tools/lib/libinterfere.so); library aggressors: the
shipped kernels at the shapes of the failing sequence-parallel test.  One JSON line per pair."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

LIB = os.path.join(ROOT, "tools", "lib", "libinterfere.so")
SRC = os.path.join(ROOT, "tools", "interfere_probe.hip")


def build():
    if not os.path.isfile(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", SRC, "-o", LIB], check=True)
    lib = C.CDLL(os.environ.get("PROBE_VICTIM_LIB") or LIB)   # e.g. tools/lib/libinterfere_nopk.so: built with -packed-fp32-ops
    lib.probe_launch.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.probe_launch.restype = C.c_int
    return lib


def main():
    if "--build-only" in sys.argv:
        build()
        return
    lib = build()
    if os.environ.get("OSK_PROBE_LIB"):   # an ablated libosk (tools/make_ablated_libs.sh) as the source of the library aggressors
        import importlib.util
        spec = importlib.util.spec_from_file_location("open_sora_amd.build", os.path.join(ROOT, "open_sora_amd", "build.py"))
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        b.LIB_PATH = os.path.join(ROOT, os.environ["OSK_PROBE_LIB"])
        b.build_lib = lambda *a, **k: b.LIB_PATH
        sys.modules["open_sora_amd.build"] = b
    from tools.xproc_probe import make_kernel
    dev = torch.device("cuda")
    main_s, side = torch.cuda.current_stream(), torch.cuda.Stream()
    report = torch.zeros(64, dtype=torch.int32, device=dev)
    K = 576
    wbuf = torch.randint(0, 2 ** 31 - 1, (4 * 64 * 16 * K // 2 + 4096,), dtype=torch.int32, device=dev)
    wbuf &= 0x3F7F3F7F   # bf16 pairs of moderate magnitude
    n16 = 1 << 16
    gbuf = ((torch.arange(n16 * 4, device=dev, dtype=torch.int64) * 2654435761) & 0xFFFFFFFF).to(torch.int64)
    gbuf = torch.where(gbuf >= 2 ** 31, gbuf - 2 ** 32, gbuf).to(torch.int32)
    dmasrc = torch.randint(0, 2 ** 31 - 1, (64 * 2048 + 256 * 512 * 4 + 4096,), dtype=torch.int32, device=dev)
    scratch = torch.zeros(16, dtype=torch.float32, device=dev)
    xg = (torch.arange(2 * K, device=dev) * 37 % 16).float() * 0.0625

    def launch(name, stream, grid, smem, p0=None, p1=None, i0=0, i1=0, i2=0):
        rc = lib.probe_launch(name.encode(), stream.cuda_stream, grid, smem, None if p0 is None else p0.data_ptr(),
                              None if p1 is None else p1.data_ptr(), i0, i1, i2)
        assert rc == 0, (name, rc)

    victims = {
        "victim_lds": lambda: launch("victim_lds", main_s, 1024, 9216, report, None, 2304, 4000),
        "victim_shfl": lambda: launch("victim_shfl", main_s, 1024, 0, report, None, 40000),
        "victim_reg": lambda: launch("victim_reg", main_s, 1024, 0, report, None, 20000),
        "victim_gld": lambda: launch("victim_gld", main_s, 1024, 0, report, gbuf, n16, 200),
        "victim_reg64": lambda: launch("victim_reg64", main_s, 1024, 0, report, None, 100000),
        "victim_pkfma": lambda: launch("victim_pkfma", main_s, 1024, 0, report, None, 400000),
        "victim_lds_long": lambda: launch("victim_lds", main_s, 1024, 9216, report, None, 2304, 24000),
        "victim_gld_long": lambda: launch("victim_gld", main_s, 1024, 0, report, gbuf, n16, 30000),
        **{f"victim_gemvvar{m}": (lambda m=m: launch("victim_gemvvar", main_s, 1024, 2 * K * 4, report, wbuf, K, 4000, m)) for m in (0, 1, 2, 3, 4, 8, 12)},
        **{f"victim_fmasrc{m}": (lambda m=m: launch("victim_fmasrc", main_s, 1024, 2 * K * 4, report, xg, K, 6000, m)) for m in (0, 1, 2, 3)},
        "victim_gemvlike": lambda: launch("victim_gemvlike", main_s, 1024, 2 * K * 4, report, wbuf, K, 4000),
    }
    lib_aggr = {}
    with torch.cuda.stream(side):
        for kind in ("gemm256p", "gemm_small", "attn"):
            lib_aggr[kind] = make_kernel(kind)

    def aggr(kind, n):
        with torch.cuda.stream(side):
            if kind == "none":
                return
            if kind in lib_aggr:
                for _ in range(n):
                    lib_aggr[kind][0]()
            elif kind == "aggr_dma_lo":
                for _ in range(n // 8):
                    launch("aggr_dma", side, 64, 16384, None, dmasrc, 2000, 0)
            elif kind == "aggr_dma_hi":
                for _ in range(n // 8):
                    launch("aggr_dma", side, 64, 98304, None, dmasrc, 2000, 90112)
            elif kind == "aggr_mfma":
                for _ in range(n // 8):
                    launch("aggr_mfma", side, 64, 0, scratch, None, 4000)
            elif kind in ("aggr_swap32", "aggr_swap16", "aggr_cvtpk", "aggr_accrw", "aggr_valu"):
                for _ in range(n // 8):
                    launch(kind, side, 64, 0, scratch, None, 20000)
            elif kind == "aggr_lds":
                for _ in range(n // 8):
                    launch("aggr_lds", side, 64, 98304, scratch, None, 98304, 40)

    aggressors = ["none", "gemm256p", "gemm_small", "attn", "aggr_dma_lo", "aggr_dma_hi", "aggr_mfma", "aggr_lds",
                  "aggr_swap32", "aggr_swap16", "aggr_cvtpk", "aggr_accrw", "aggr_valu"]
    only_a = [a.split("=")[1] for a in sys.argv if a.startswith("--aggressors=")]
    if only_a:
        aggressors = only_a[0].split(",")
    only_v = [a.split("=")[1] for a in sys.argv if a.startswith("--victims=")]
    vs = only_v[0].split(",") if only_v else list(victims) + ["osk_gemv"]
    gemv = make_kernel("gemv")
    for v in vs:
        for a in aggressors:
            report.zero_()
            torch.cuda.synchronize()
            aggr(a, 1500)
            if v == "osk_gemv":
                gemv[0]()
                torch.cuda.synchronize()
                first = gemv[1]()[0].clone()
                aggr(a, 3000)
                bad = torch.zeros((), dtype=torch.int64, device=dev)
                for _ in range(300):
                    gemv[1]()[0].zero_()
                    gemv[0]()
                    bad += (gemv[1]()[0] != first).any()
                torch.cuda.synchronize()
                print(json.dumps(dict(victim=v, aggressor=a, bad_launches=int(bad), of=300)), flush=True)
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_s)
            victims[v]()
            e1.record(main_s)
            torch.cuda.synchronize()
            r = report.cpu().tolist()
            recs = [[hex(x & 0xFFFFFFFF) for x in r[4 + 4 * i: 8 + 4 * i]] for i in range(min(r[0], 6))]
            print(json.dumps(dict(victim=v, aggressor=a, mismatches=r[0], victim_ms=round(e0.elapsed_time(e1), 2), records=recs)), flush=True)


if __name__ == "__main__":
    main()
