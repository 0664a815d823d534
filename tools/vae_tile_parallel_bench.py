#!/usr/bin/env python
"""Tiled VAE decode with the tiles spread over the ranks of a node (AutoencoderKLCausal3D.enable_tile_parallel; SURVEY.md 8(e) "VAE",
8(f) rank 2; the reference's tiled decode: /root/reference/opensora/models/hunyuan_vae/autoencoder_kl_causal_3d.py:384-552).
One process per GPU (torchrun, RCCL); world 1 = the serial tiled decode.  Prints one JSON line on rank 0: ms per decode (max over
ranks) and a checksum of the decoded video -- the same on every world size (tile parallelism is bit-identical by construction).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/vae_tile_parallel_bench.py
"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch


def main():
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    from open_sora_amd import configs, hunyuan_vae

    T_lat, h, w = (int(v) for v in os.environ.get("OSK_VAE_LATENT", "9,90,160").split(","))   # 33 frames of 720p
    torch.manual_seed(1234)
    model = hunyuan_vae.CausalVAE3D_HUNYUAN(device_map=dev, torch_dtype=torch.bfloat16, **configs.VAE["hunyuan"])
    model.enable_tiling()
    if world > 1:
        model.enable_tile_parallel()
    g = torch.Generator(device=dev).manual_seed(42)
    z = torch.randn(1, 16, T_lat, h, w, device=dev, generator=g).to(torch.bfloat16)
    with torch.inference_mode():
        out = model.decode(z)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        n = 3
        t0 = time.perf_counter()
        for _ in range(n):
            out = model.decode(z)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        ms = (time.perf_counter() - t0) / n * 1e3
    if dist is not None:
        t = torch.tensor([ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
    digest = hashlib.sha256(out.float().cpu().numpy().tobytes()).hexdigest()[:16]
    if rank == 0:
        print(json.dumps({"what": "tiled VAE decode, tiles spread over ranks", "n_gpus": world, "latent": [T_lat, h, w], "out_shape": list(out.shape),
                          "ms_per_decode": round(ms, 2), "sha256_16": digest}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
