"""GPU, RCCL (`nccl` backend), one process per GPU: the sequence-parallel denoise step on the HIP kernels (K / V^T
all-gather and the head-exchanging all-to-all, open_sora_amd/seqpar.py) against the single-GPU forward of the same
process and the oracle.  Needs >= 2 visible GPUs; on a 1-GPU box it is SKIPPED (the gloo tests of
tests/test_seqpar_gloo.py then remain the only coverage of the exchange logic).
Reference: mmdit_model_forward, /root/reference/opensora/models/mmdit/distributed.py:580-683 (ring :223-313, Ulysses :473-495)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _ngpu():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, geom, q, mode, fp8):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from open_sora_amd import mmdit, seqpar
        from oracle import configs
        from tests.util import torch_inputs, torch_params

        cfg = configs.GOLDEN[name][0]
        B, T, h, w, L_txt = geom
        model = mmdit.Flux(device_map=dev, torch_dtype=torch.bfloat16, **cfg)
        model.load_state_dict(torch_params(cfg, dtype=torch.bfloat16, device=dev), strict=True)
        if fp8:
            model.enable_fp8()
        inp = torch_inputs(cfg, B, T, h, w, L_txt, dtype=torch.bfloat16, device=dev)
        with torch.inference_mode():
            single = model(**inp).float().cpu()
            sp = seqpar.enable(model, mode=mode)
            assert model._sp is not None and sp.P == world and dist.get_backend() == "nccl"
            # 12 forwards back to back, no host sync in between: repeated calls reuse the exchange buffers while RCCL's kernels share
            # the CUs with the hand-scheduled MFMA loops -- the screen that caught the packed-FP32 cross-kernel interference on one GPU
            # (profiles/r03_cross_kernel_interference.md) applied to the first real multi-GPU run
            outs_dev = [model(**inp) for _ in range(12)]
            torch.cuda.synchronize()
            outs = [o.float().cpu() for o in outs_dev]
            sp.exposed = []                                           # exposed-communication accounting (bench.py --gpus N)
            model(**inp)
            torch.cuda.synchronize()
            summary = sp.exposed_summary()
            assert summary and all(v >= 0.0 for v in summary.values()) and "out" in summary, summary
            sp.exposed = None
            seqpar.disable(model)
        torch.cuda.synchronize()
        bad = [i for i, o in enumerate(outs[1:], 1) if not torch.equal(outs[0], o)]
        assert not bad, f"sequence-parallel forward is not repeatable: forwards {bad} of 12 differ from forward 0"
        q.put((rank, single.numpy(), outs[0].numpy()))
    except BaseException:
        import traceback

        q.put((rank, "error", traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def _run(world, name, geom, mode, fp8=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, geom, q, mode, fp8)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    for _ in range(world):
        r = q.get(timeout=600)
        if isinstance(r[1], str):
            for p in procs:
                p.kill()
            pytest.fail(f"rank {r[0]} failed:\n{r[2]}")
        res.append(r)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


# (world, golden config, (B, T, h, w, L_txt)): L divisible by world, L / world > L_txt; token counts large enough for
# whole and ragged 64-key tiles per segment
CASES = [
    (2, "hd72_eager_split", (2, 4, 8, 8, 64)),     # L = 320, 160 per rank
    (2, "hd128_liger_split", (3, 2, 9, 7, 22)),    # L = 148, 74 per rank (ragged), CFG-triple batch
    (2, "hd64_eager_fused", (1, 3, 8, 8, 64)),     # L = 256
    (4, "hd72_eager_split", (2, 4, 8, 8, 64)),     # 8 heads / 4 ranks, 80 per rank
    (8, "hd72_eager_split", (1, 8, 8, 8, 64)),     # 8 heads / 8 ranks: L = 576, 72 per rank
]


@pytest.mark.parametrize("mode", ["allgather", "ulysses"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"w{c[0]}-{c[1]}")
def test_seqpar_nccl_matches_single_gpu_and_oracle(hip_lib, case, mode):
    world, name, geom = case
    if _ngpu() < world:
        pytest.skip(f"needs {world} GPUs, {_ngpu()} visible")
    from oracle import configs, mmdit_oracle as O
    from tests.util import rel_l2, torch_inputs, torch_params

    cfg = configs.GOLDEN[name][0]
    if mode == "ulysses" and cfg["num_heads"] % world:
        pytest.skip("head exchange needs num_heads % world == 0")
    B, T, h, w, L_txt = geom
    L = L_txt + T * h * w
    assert L % world == 0 and L // world > L_txt
    res = _run(world, name, geom, mode)
    with torch.inference_mode():
        truth = O.forward(torch_params(cfg), cfg, **torch_inputs(cfg, B, T, h, w, L_txt))
        ref_bf16 = O.forward(torch_params(cfg, dtype=torch.bfloat16), cfg,
                             **torch_inputs(cfg, B, T, h, w, L_txt, dtype=torch.bfloat16))
    e_ref = rel_l2(ref_bf16.float(), truth)
    for rank, single, sharded in res:
        single, sharded = torch.from_numpy(single), torch.from_numpy(sharded)
        e1, eP = rel_l2(single, truth), rel_l2(sharded, truth)
        print(f"rank {rank}: relL2 single-GPU {e1:.3e} sequence-parallel {eP:.3e} ref-bf16 {e_ref:.3e}")
        assert eP <= max(1.5 * e_ref, 2.0 ** -8), (rank, eP, e_ref)
        assert rel_l2(sharded, single) <= 2.0 ** -7, (rank, rel_l2(sharded, single))
    for rank, _, sharded in res[1:]:
        assert np.array_equal(sharded, res[0][2]), "ranks disagree on the gathered prediction"


@pytest.mark.parametrize("mode", ["allgather", "ulysses"])
def test_seqpar_nccl_fp8_mode(hip_lib, mode):
    if _ngpu() < 2:
        pytest.skip(f"needs 2 GPUs, {_ngpu()} visible")
    from tests.util import rel_l2

    res = _run(2, "hd72_eager_split", (2, 4, 8, 8, 64), mode, fp8=True)
    for rank, single, sharded in res:
        assert rel_l2(torch.from_numpy(sharded), torch.from_numpy(single)) <= 2e-2
    assert np.array_equal(res[1][2], res[0][2])
