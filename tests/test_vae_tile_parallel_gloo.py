"""CPU, world_size 2 over gloo: tile-level parallelism of the tiled VAE encode / decode (open_sora_amd/hunyuan_vae.py
::enable_tile_parallel -- the tiles of the reference's own tiling loops, autoencoder_kl_causal_3d.py:384-552, are the
independent units; results are exchanged by one broadcast per tile, blends and assembly are replicated).  Every rank
must return exactly what the single-process tiled path returns.  Kernels = the CPU emulation (tests/cpu_ops.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, spatial, temporal):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from open_sora_amd import hunyuan_vae, mmdit
        from oracle import configs, synth
        from tests import cpu_ops

        mmdit.set_ops_for_testing(cpu_ops)
        cfg, B, T, H, W = configs.VAE_TILED_GOLDEN["c32_tiled"]
        sd = {k: torch.from_numpy(v).bfloat16() for k, v in synth.make_params(synth.vae_param_shapes(cfg), 0).items()}
        m = hunyuan_vae.CausalVAE3D_HUNYUAN(device_map="cpu", torch_dtype=torch.bfloat16, **cfg)
        m.load_state_dict(sd, strict=True)
        m.enable_spatial_tiling(spatial)
        m.enable_temporal_tiling(temporal)
        x = torch.from_numpy(synth.vae_video(B, T, H, W)).bfloat16()
        zin = torch.from_numpy(synth.vae_latent(B, (T - 1) // 4 + 1, H // 8, W // 8)).bfloat16()
        with torch.inference_mode():
            z1 = m.encode(x, sample_posterior=False).float().clone()
            d1 = m.decode(zin).float().clone()
            m.enable_tile_parallel()
            z2 = m.encode(x, sample_posterior=False).float().clone()
            d2 = m.decode(zin).float().clone()
            m.disable_tile_parallel()
        q.put((rank, z1.numpy(), d1.numpy(), z2.numpy(), d2.numpy()))
    except BaseException:
        import traceback

        q.put((rank, "error", traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("spatial,temporal", [(True, True), (True, False), (False, True)])
def test_tile_parallel_matches_single_process(spatial, temporal):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, spatial, temporal)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    for _ in range(world):
        r = q.get(timeout=900)
        if isinstance(r[1], str):
            for p in procs:
                p.kill()
            pytest.fail(f"rank {r[0]} failed:\n{r[2]}")
        res.append(r)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, z1, d1, z2, d2 in res:
        assert np.array_equal(z1, z2), f"rank {rank}: tile-parallel encode differs from the single-process tiled encode"
        assert np.array_equal(d1, d2), f"rank {rank}: tile-parallel decode differs from the single-process tiled decode"
    assert np.array_equal(res[0][4], res[1][4]) and np.array_equal(res[0][3], res[1][3])
