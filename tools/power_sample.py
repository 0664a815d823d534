#!/usr/bin/env python
"""Socket power, clocks and throughput of the MFMA kernels under sustained load (one arm at a time, ~4 s each, random bf16
data): does the board hold POWER constant while busy x clock differs per kernel?

  arms: this library's GEMM and hipBLASLt at 8192^3 and at the MLP-down shape, the attention kernel at the bench shape.
While an arm loops on the GPU, a sampler thread polls `rocm-smi --showpower -c --json` every ~150 ms.
Output: one JSON line per arm (gpurun_out/power_sample.jsonl)."""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from tools.ab_vendor import attn_arms, gemm_arms, ATTN_SHAPE


def poll(stop, samples):
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "-c", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            card = next(iter(d.values()))
            samples.append({k: v for k, v in card.items() if "ower" in k or "sclk" in k or "mclk" in k or "fclk" in k})
        except Exception as e:  # noqa: BLE001
            samples.append({"error": str(e)[:80]})
        time.sleep(0.15)


def run_arm(name, fn, flops, seconds=4.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, samples))
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        n += 10
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    keep = samples[len(samples) // 3:]          # steady state: drop the first third
    def num(s):
        try:
            return float(str(s).strip("()").replace("Mhz", "").replace("MHz", "").split()[0])
        except Exception:  # noqa: BLE001
            return None
    agg = {}
    for s in keep:
        for k, v in s.items():
            x = num(v)
            if x is not None:
                agg.setdefault(k, []).append(x)
    rec = {"arm": name, "tflops": round(flops * n / dt / 1e12, 1), "launches": n, "samples": len(keep),
           **{k: round(sum(v) / len(v), 1) for k, v in agg.items()}}
    print(json.dumps(rec), flush=True)
    return rec


def main():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    recs = []
    for (M, N, K) in ((8192, 8192, 8192), (50688, 1152, 4608), (50688, 4608, 1152)):
        arms, _keep = gemm_arms(M, N, K)
        for a in ("osk", "vendor"):
            recs.append(run_arm(f"gemm {M}x{N}x{K} {a}", arms[a], 2.0 * M * N * K))
        del arms, _keep
        torch.cuda.empty_cache()
    B, H, L, hd = ATTN_SHAPE
    arms, _keep = attn_arms(B, H, L, hd)
    recs.append(run_arm("attention osk (attn_asm72)", arms["osk"], 4.0 * B * H * L * L * hd))
    with open(os.path.join(ROOT, "gpurun_out", "power_sample.jsonl"), "w") as f:
        for r in recs:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
