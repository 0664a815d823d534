#!/usr/bin/env python
"""Do the fp8 Linear kernels repeat bit for bit while ANOTHER process hammers the same GPU?  (tests/test_gpu_seqpar_1gpu.py's
fp8 cases: two ranks on one GPU, sporadic 1-ulp differences in batch 0.)  Two processes, each: quantize_rows_fp8 + gemm_fp8 and
ln_modulate_fp8 on fixed inputs, N iterations, every result compared with the first."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.multiprocessing as mp


def worker(rank, iters):
    torch.cuda.set_device(0)
    from open_sora_amd import _C
    BF = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(3)
    res = {}
    for (M, N, K) in ((256, 576, 2304), (288, 576, 2304), (256, 2304, 2304)):
        a = torch.randn(1, M, K, device="cuda", generator=g).to(BF)
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(BF)
        bias = torch.randn(N, device="cuda", generator=g).float()
        w8, sw = _C.quantize_rows_fp8(w)
        out = torch.empty(1, M, N, dtype=BF, device="cuda")
        first_q = first_o = None
        bad_q = bad_o = 0
        for i in range(iters):
            a8, sa = _C.quantize_rows_fp8(a)
            _C.gemm_fp8(a8, sa, w8, sw, bias, out)
            torch.cuda.synchronize()
            q = (a8.view(torch.uint8).clone(), sa.clone())
            o = out.clone()
            if first_q is None:
                first_q, first_o = q, o
            else:
                bad_q += int(not (torch.equal(q[0], first_q[0]) and torch.equal(q[1], first_q[1])))
                bad_o += int(not torch.equal(o, first_o))
        res[(M, N, K)] = (bad_q, bad_o)
    print(f"rank {rank}: (quantize mismatches, gemm_fp8 mismatches) of {iters - 1}: {res}", flush=True)
    # the MLP of a block as the model runs it: bf16 GEMM + GELU -> quantize -> fp8 GEMM with gate * x + residual IN PLACE, no
    # synchronisation in between, [2, 144, 576] activations (M = 288: the batch boundary cuts through the first 256-row tile, the second one is ragged)
    B, L, D, F = 2, 144, 576, 2304
    x0 = torch.randn(B, L, D, device="cuda", generator=g).to(BF)
    xm = torch.randn(B, L, D, device="cuda", generator=g).to(BF)
    w1 = (torch.randn(F, D, device="cuda", generator=g) * D ** -0.5).to(BF)
    w2 = (torch.randn(D, F, device="cuda", generator=g) * F ** -0.5).to(BF)
    b1, b2 = torch.randn(F, device="cuda", generator=g).float(), torch.randn(D, device="cuda", generator=g).float()
    gate = torch.randn(B, D, device="cuda", generator=g).float()
    w28, sw2 = _C.quantize_rows_fp8(w2)
    h = torch.empty(B, L, F, dtype=BF, device="cuda")
    first, bad = None, 0
    for i in range(iters):
        x = x0.clone()
        _C.gemm(xm, w1, b1, h, gelu_from=0)
        a8, sa = _C.quantize_rows_fp8(h)
        _C.gemm_fp8(a8, sa, w28, sw2, b2, x, res=x, gate=gate, gate_batch_stride=gate.stride(0))
        if i % 8 == 7:
            torch.cuda.synchronize()
        o = x.clone()
        if first is None:
            first = o
        else:
            bad += int(not torch.equal(o, first))
    torch.cuda.synchronize()
    print(f"rank {rank}: MLP chain (gemm + GELU -> quantize -> gemm_fp8 gate/res in place) mismatches of {iters - 1}: {bad}", flush=True)


if __name__ == "__main__":
    mp.set_start_method("spawn")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ps = [mp.Process(target=worker, args=(r, 300)) for r in range(n)]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
