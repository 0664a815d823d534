"""CPU, world_size 2 and 4 over gloo: the sequence-parallel denoise step (open_sora_amd/seqpar.py — token sharding,
K/V all-gather in the segment layout the attention kernel addresses, output gather) must reproduce the
single-process result and the reference golden.  Kernels are the CPU emulation of their semantics
(tests/cpu_ops.py); the RCCL/HIP path runs the same host code."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, geom, q, mode, fp8=False):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from open_sora_amd import mmdit, seqpar
        from oracle import configs
        from tests import cpu_ops
        from tests.util import torch_inputs, torch_params

        mmdit.set_ops_for_testing(cpu_ops)
        cfg = configs.GOLDEN[name][0]
        B, T, h, w, L_txt = geom
        model = mmdit.Flux(device_map="cpu", torch_dtype=torch.bfloat16, **cfg)
        model.load_state_dict(torch_params(cfg, dtype=torch.bfloat16), strict=True)
        if fp8:   # fp8 mode: attention with the fp8 P.V product (the tiny Linears of these geometries stay bf16)
            model.enable_fp8()
        inp = torch_inputs(cfg, B, T, h, w, L_txt, dtype=torch.bfloat16)
        with torch.inference_mode():
            single = model(**inp).float().clone()
            sp = seqpar.enable(model, mode=mode)
            assert sp.head_parallel(cfg["num_heads"]) == (mode == "ulysses")
            assert model._sp is not None and sp.P == world
            sharded = model(**inp).float().clone()
            seqpar.disable(model)
        q.put((rank, single.numpy(), sharded.numpy()))
    except BaseException as e:  # surface the failure instead of letting the parent wait for the queue timeout
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
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


# (world, golden config name, (B, T, h, w, L_txt) in token-grid units): L = L_txt + T*h*w divisible by world and
# L/world > L_txt (rank 0 holds all text rows plus some image rows, the other ranks image rows only)
CASES = [
    (2, "hd64_eager_fused", (2, 2, 2, 3, 4)),     # L = 16, 8 per rank
    (4, "hd72_eager_split", (2, 2, 3, 5, 2)),     # L = 32, 8 per rank
    (2, "hd128_liger_split", (3, 1, 2, 3, 2)),    # L = 8, 4 per rank, CFG-triple batch
    (2, "hd72_liger_fused", (1, 5, 3, 7, 15)),    # L = 120, 60 per rank: ragged 64-key segment tiles
]


# head-parallel ("ulysses") exchange needs num_heads % world == 0: 2 heads / 2 ranks, 8 heads / 4 and 2 ranks
MODE_CASES = [(c, "allgather") for c in CASES] + [(CASES[0], "ulysses"), (CASES[1], "ulysses"), (CASES[2], "ulysses"),
                                                    (CASES[3], "ulysses")]


@pytest.mark.parametrize("case,mode", MODE_CASES, ids=lambda v: v if isinstance(v, str) else f"w{v[0]}-{v[1]}")
def test_seqpar_matches_single_process_and_oracle(case, mode):
    world, name, geom = case
    from oracle import configs, mmdit_oracle as O
    from tests.util import rel_l2, torch_inputs, torch_params

    cfg = configs.GOLDEN[name][0]
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
        print(f"rank {rank}: relL2 single {e1:.3e} sharded {eP:.3e} ref-bf16 {e_ref:.3e}")
        assert eP <= max(1.5 * e_ref, 2.0 ** -8), (rank, eP, e_ref)
        # same kernels and per-row arithmetic; only the key order inside the softmax sums differs
        assert rel_l2(sharded, single) <= 2.0 ** -7, (rank, rel_l2(sharded, single))
    for rank, _, sharded in res[1:]:  # every rank returns the same full prediction
        assert np.array_equal(sharded, res[0][2])


@pytest.mark.parametrize("case,mode", [(CASES[2], "allgather"), (CASES[3], "allgather"), (CASES[1], "ulysses"), (CASES[3], "ulysses")],
                         ids=lambda v: v if isinstance(v, str) else f"w{v[0]}-{v[1]}")
def test_seqpar_fp8_mode(case, mode):
    """fp8 mode under sequence parallelism: V travels / is re-laid out as e4m3 V^T, its per-(batch, head) scale agreed
    across ranks (all-reduce max when K / V^T are gathered; local when heads are exchanged), attention with the fp8 P.V
    product over P key segments.  Sharded == single-process fp8 mode up to the e4m3 rounding of P against different
    reference maxima; every rank returns the same prediction."""
    world, name, geom = case
    from tests.util import rel_l2

    res = _run(world, name, geom, mode, fp8=True)
    for rank, single, sharded in res:
        single, sharded = torch.from_numpy(single), torch.from_numpy(sharded)
        assert rel_l2(sharded, single) <= 2e-2, (rank, rel_l2(sharded, single))
    for rank, _, sharded in res[1:]:
        assert np.array_equal(sharded, res[0][2])
