"""CPU: the sequence-parallel step with P ranks as THREADS of one process over tests/local_transport.py (the transport the GPU
overlap test uses, here without streams) and the CPU emulation of the kernels: the transport abstraction of
open_sora_amd.seqpar (no torch.distributed involved), the thread protocol, shared plan / private workspaces."""
import copy

import pytest
import torch

from oracle import configs
from tests import cpu_ops
from tests.local_transport import LocalTransport, run_ranks
from tests.util import rel_l2, torch_inputs, torch_params

BF = torch.bfloat16


@pytest.fixture()
def cpu_mmdit(hip_lib):
    from open_sora_amd import mmdit

    mmdit.set_ops_for_testing(cpu_ops)
    yield mmdit
    mmdit.set_ops_for_testing(hip_lib)


@pytest.mark.parametrize("mode", ["allgather", "ulysses"])
@pytest.mark.parametrize("case", [(2, "hd64_eager_fused", (2, 2, 2, 3, 4)), (4, "hd72_eager_split", (2, 2, 3, 5, 2))], ids=lambda c: f"w{c[0]}-{c[1]}")
def test_ranks_as_threads_match_single_process(cpu_mmdit, case, mode):
    from open_sora_amd import seqpar

    P, name, geom = case
    cfg = configs.GOLDEN[name][0]
    B, T, h, w, L_txt = geom
    model = cpu_mmdit.Flux(device_map="cpu", torch_dtype=BF, **cfg)
    model.load_state_dict(torch_params(cfg, dtype=BF), strict=True)
    inp = torch_inputs(cfg, B, T, h, w, L_txt, dtype=BF)
    with torch.inference_mode():
        single = model(**inp).float()

    def rank_fn(rank, world):
        m = copy.copy(model)
        m.forward = m.forward_ckpt
        object.__setattr__(m, "_osk_ws_cache", {})
        tp = LocalTransport(world, rank, "cpu")
        sp = seqpar.enable(m, mode=mode, transport=tp)
        assert sp.P == P and sp.rank == rank and sp.head_parallel(cfg["num_heads"]) == (mode == "ulysses")
        with torch.inference_mode():
            return [m(**inp).float().clone() for _ in range(2)], tp.calls

    res = run_ranks(P, rank_fn, "cpu")
    for outs, calls in res:
        assert calls > 0 and torch.equal(outs[0], outs[1]) and torch.equal(outs[0], res[0][0][0])
    assert rel_l2(res[0][0][0], single) <= 2.0 ** -7


def test_allgather_ranks_write_v_straight_into_the_gather_slot(cpu_mmdit):
    """round 6: in all-gather mode a rank's K / V projection is one osk_gemm_group_bf16 call whose V^T task writes into the rank's slot
    of the gathered V^T buffer -- no osk_v_transpose_bf16 pass on the rank (shapes large enough for the 256 x 256 tile path: B = 4);
    same result as the single-process forward."""
    from open_sora_amd import seqpar

    P, name = 2, "hd72_eager_split"
    cfg = configs.GOLDEN[name][0]
    B, T, h, w, L_txt = 4, 4, 8, 8, 64                     # L = 256 + 64 = 320: 160 rows per rank (rank 0: 64 txt + 96 img)
    model = cpu_mmdit.Flux(device_map="cpu", torch_dtype=BF, **cfg)
    model.load_state_dict(torch_params(cfg, dtype=BF), strict=True)
    inp = torch_inputs(cfg, B, T, h, w, L_txt, dtype=BF)
    with torch.inference_mode():
        single = model(**inp).float()
    calls = {"vt": 0, "group": 0}
    real_vt, real_group = cpu_ops.v_transpose, cpu_ops.gemm_group
    import threading
    in_group = threading.local()

    def spy_vt(*a, **k):
        if not getattr(in_group, "on", False):            # (the emulation's own V^T task calls v_transpose: not the engine's pass)
            calls["vt"] += 1
        return real_vt(*a, **k)

    def spy_group(tasks):
        in_group.on = True
        try:
            ok = real_group(tasks)
        finally:
            in_group.on = False
        calls["group"] += bool(ok)
        return ok

    def rank_fn(rank, world):
        m = copy.copy(model)
        m.forward = m.forward_ckpt
        object.__setattr__(m, "_osk_ws_cache", {})
        sp = seqpar.enable(m, mode="allgather", transport=LocalTransport(world, rank, "cpu"))
        assert not sp.head_parallel(cfg["num_heads"])
        with torch.inference_mode():
            return m(**inp).float().clone()

    cpu_ops.v_transpose, cpu_ops.gemm_group = spy_vt, spy_group
    try:
        res = run_ranks(P, rank_fn, "cpu")
    finally:
        cpu_ops.v_transpose, cpu_ops.gemm_group = real_vt, real_group
    n_blocks = cfg["depth"] + cfg["depth_single_blocks"]
    assert calls["group"] == P * n_blocks and calls["vt"] == 0, calls
    assert torch.equal(res[0], res[1]) and rel_l2(res[0], single) <= 2.0 ** -7
