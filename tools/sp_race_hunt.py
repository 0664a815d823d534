#!/usr/bin/env python
"""Repeatability hunt for the sequence-parallel forward (VERDICT r2 lead item: 4 back-to-back forwards of the same inputs
were not bit-identical on the driver's box, two ranks sharing one GPU).

World-size P ranks as processes on ONE GPU (collectives staged through host memory over gloo, exactly as
tests/test_gpu_seqpar_1gpu.py), optional contender processes / a contender stream, N forwards per rank:

  * every forward's prediction is compared with the first one's;
  * --instrument: every kernel-table call (mmdit.ops()) and every collective is followed by a device-side checksum of each
    tensor argument, enqueued on the SAME stream (no host synchronisation is added, so the timing of the launch sequence
    stays close to the plain run); after the forward the checksum chains of run i and run 0 are diffed: the first entry that
    differs names the kernel, the argument and whether its inputs still agreed;
  * --poison nan|rand: every workspace the forward re-uses (model, sequence-parallel buffers, attention tail workspace) and
    every torch.empty() it makes is pre-filled with NaN / per-run random bits: a read of memory the forward did not write
    first then shows up deterministically (NaN: loudly), with no contention needed.

    python tools/sp_race_hunt.py --runs 50 --hammer matmul,copy --instrument --out gpurun_out/race/a.json
"""
import argparse
import json
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.multiprocessing as mp

WRAP = {"ln_modulate", "gemm", "gemv_tasks", "timestep_embedding", "rope_table", "qknorm_rope", "v_transpose", "attention_fwd",
        "ln_modulate_fp8", "quantize_rows_fp8", "gemm_fp8", "v_scale_fp8", "v_transpose_fp8", "attention_fwd_pv8", "cfg_euler"}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_W = {}


def dev_hash(t: torch.Tensor) -> torch.Tensor:
    """position-weighted byte checksum of a (strided) device tensor, computed on the current stream; int64 scalar on device"""
    x = t.detach().contiguous().reshape(-1).view(torch.uint8)
    n = x.numel()
    if n == 0:
        return torch.zeros((), dtype=torch.int64, device=t.device)
    w = _W.get((n, str(t.device)))
    if w is None:
        w = _W[(n, str(t.device))] = (torch.arange(n, device=t.device, dtype=torch.int32) % 8191 + 1)
    return (x.to(torch.int32) * w).to(torch.int64).sum()


class HashOps:
    """kernel table proxy: checksums of every tensor argument before and after each call, kept on the device"""

    def __init__(self, inner, max_bytes=1 << 26):
        self._inner = inner
        self.names, self.hashes = [], []
        self.max_bytes = max_bytes
        self.enabled = True

    def reset(self):
        self.names, self.hashes = [], []

    def mark(self, label, tensors):
        if not self.enabled:
            return
        for i, t in enumerate(tensors):
            self.names.append((label, i, "post"))
            self.hashes.append(dev_hash(t))

    def __getattr__(self, name):
        fn = getattr(self._inner, name)
        if name not in WRAP:
            return fn

        def wrapped(*a, **k):
            if not self.enabled:
                return fn(*a, **k)
            args = list(enumerate(a)) + list(k.items())
            tens = [(i, t) for i, t in args if isinstance(t, torch.Tensor) and t.is_cuda and t.numel() * t.element_size() <= self.max_bytes
                    and i != "workspace"]
            idx = len([n for n in self.names if n[2] == "call"])
            for i, t in tens:
                self.names.append((f"{idx}:{name}", i, "pre"))
                self.hashes.append(dev_hash(t))
            r = fn(*a, **k)
            self.names.append((f"{idx}:{name}", None, "call"))
            self.hashes.append(torch.zeros((), dtype=torch.int64, device="cuda"))
            for i, t in tens:
                self.names.append((f"{idx}:{name}", i, "post"))
                self.hashes.append(dev_hash(t))
            rets = r if isinstance(r, (tuple, list)) else (r,)
            for j, t in enumerate(rets):
                if isinstance(t, torch.Tensor) and t.is_cuda and not any(t is u for _, u in tens):
                    self.names.append((f"{idx}:{name}", f"ret{j}", "post"))
                    self.hashes.append(dev_hash(t))
            return r

        return wrapped

    def collect(self):
        h = torch.stack(self.hashes).cpu().tolist() if self.hashes else []
        return list(self.names), h


class _Done:
    def wait(self):
        return True


def stage_collectives_through_host(dist, hooks):
    real_ag, real_a2a, real_ar = dist.all_gather_into_tensor, dist.all_to_all_single, dist.all_reduce

    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        torch.cuda.synchronize()
        o = torch.empty(out.shape, dtype=out.dtype, device="cpu")
        real_ag(o, inp.cpu(), group=group)
        out.copy_(o)
        hooks("all_gather", [out])
        return _Done() if async_op else None

    def all_to_all_single(out, inp, group=None, async_op=False):
        torch.cuda.synchronize()
        o = torch.empty(out.shape, dtype=out.dtype, device="cpu")
        real_a2a(o, inp.cpu(), group=group)
        out.copy_(o)
        hooks("all_to_all", [out])
        return _Done() if async_op else None

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        torch.cuda.synchronize()
        c = t.cpu()
        real_ar(c, op=op, group=group)
        t.copy_(c)
        hooks("all_reduce", [t])
        return _Done() if async_op else None

    dist.all_gather_into_tensor, dist.all_to_all_single, dist.all_reduce = all_gather_into_tensor, all_to_all_single, all_reduce


def _poison_tensor(t, kind, gen):
    if kind == "nan":
        if t.dtype in (torch.bfloat16, torch.float32, torch.float64, torch.float16):
            t.fill_(float("nan"))
        elif t.dtype == torch.uint8:
            t.fill_(0x7F)   # e4m3 NaN
    elif kind == "rand":
        raw = t.view(torch.uint8) if t.is_contiguous() else None
        if raw is not None:
            raw.copy_(torch.randint(0, 256, raw.shape, dtype=torch.uint8, device=t.device, generator=gen))
            if t.dtype == torch.bfloat16:   # keep the garbage finite and moderate: clear the top exponent bit
                t.view(torch.int16).bitwise_and_(0x3FFF | -0x8000)
            elif t.dtype == torch.float32:
                t.view(torch.int32).bitwise_and_(0x3FFFFFFF | -0x80000000)


def install_empty_poison(kind, gen_holder):
    real_empty, real_empty_like = torch.empty, torch.empty_like

    def empty(*a, **k):
        t = real_empty(*a, **k)
        if t.is_cuda and gen_holder.get("on"):
            _poison_tensor(t, kind, gen_holder.get("gen"))
        return t

    def empty_like(*a, **k):
        t = real_empty_like(*a, **k)
        if t.is_cuda and gen_holder.get("on"):
            _poison_tensor(t, kind, gen_holder.get("gen"))
        return t

    torch.empty, torch.empty_like = empty, empty_like


def poison_workspaces(model, sp, kind, gen, _C):
    for ws in getattr(model, "_osk_ws_cache", {}).values():
        for name in ("x", "xm", "y", "h", "vt"):
            _poison_tensor(getattr(ws, name), kind, gen)
        if getattr(ws, "vt8", None) is not None:
            _poison_tensor(ws.vt8, kind, gen)
    if sp is not None:
        for b in sp._bufs.values():
            for t in (b.values() if isinstance(b, dict) else b):
                if isinstance(t, torch.Tensor):
                    _poison_tensor(t, kind, gen)
    for t in _C._ATTN_WS.values():
        _poison_tensor(t, kind, gen)


def describe_diff(a: torch.Tensor, b: torch.Tensor, P: int, L_txt: int):
    d = (a.float() - b.float()).abs()
    bad = (d > 0) | (torch.isnan(a.float()) != torch.isnan(b.float()))
    nz = bad.nonzero()
    B, L_img, C = a.shape
    Lloc = (L_img + L_txt) // P
    rows = sorted(set(nz[:, 1].tolist()))
    owners = sorted({(r + L_txt) // Lloc for r in rows})
    return dict(n=int(bad.sum()), numel=a.numel(), max=float(torch.nan_to_num(d, nan=1e30).max()), max_out=float(a.float().abs().max()),
                batches=sorted(set(nz[:, 0].tolist())), row_min=rows[0], row_max=rows[-1], n_rows=len(rows), owner_ranks=owners,
                finite=bool(torch.isfinite(b.float()).all()))


def hammer(kind, stop_at, dev_stream=None):
    """contender loop until time.time() > stop_at (or forever when stop_at is None; the parent kills the process)"""
    torch.cuda.set_device(0)
    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.randn(4096, 4096, device="cuda", generator=g).bfloat16()
    b = torch.randn(4096, 4096, device="cuda", generator=g).bfloat16()
    src = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    small = torch.randn(1 << 16, device="cuda", generator=g)
    i = 0
    while stop_at is None or time.time() < stop_at:
        if kind == "matmul":
            for _ in range(8):
                a @ b
        elif kind == "copy":
            for _ in range(8):
                dst.copy_(src)
        else:   # "small": many tiny low-occupancy launches that can share a CU with anything
            for _ in range(64):
                small.mul_(1.0000001)
        i += 1
        if i % 16 == 0:
            torch.cuda.synchronize()


def hammer_proc(kind):
    hammer(kind, None)


def worker(rank, args, port, q):
    import faulthandler
    import torch.distributed as dist

    faulthandler.enable()   # a GPU memory fault aborts the process: with HIP_LAUNCH_BLOCKING=1 the Python stack names the launch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=args.world)
    try:
        torch.cuda.set_device(0)
        torch.set_num_threads(4)
        from open_sora_amd import _C, mmdit, seqpar
        from oracle import configs
        from tests.util import torch_inputs, torch_params

        hops = HashOps(_C)
        hops.enabled = False
        if args.instrument:
            mmdit.set_ops_for_testing(hops)
        if args.checkpoints:   # light instrumentation: the residual stream after every block (+ the gathered buffers, below)
            rd, rs = mmdit.run_double_block, mmdit.run_single_block
            counter = [0]

            def run_double_block(plan, ws, *a, **k):
                r = rd(plan, ws, *a, **k)
                hops.mark(f"after double block (call {counter[0]})", [ws.x])
                counter[0] += 1
                return r

            def run_single_block(plan, ws, *a, **k):
                r = rs(plan, ws, *a, **k)
                hops.mark(f"after single block (call {counter[0]})", [ws.x])
                counter[0] += 1
                return r

            mmdit.run_double_block, mmdit.run_single_block = run_double_block, run_single_block
        stage_collectives_through_host(dist, hops.mark)
        gen_holder = {"on": False, "gen": torch.Generator(device="cuda")}
        if args.poison != "none":
            install_empty_poison(args.poison, gen_holder)
        cfg = dict(configs.GOLDEN[args.name][0])
        if args.depth is not None:
            cfg["depth"], cfg["depth_single_blocks"] = args.depth
        B, T, h, w, L_txt = args.geom
        model = mmdit.Flux(device_map="cuda:0", torch_dtype=torch.bfloat16, **cfg)
        model.load_state_dict(torch_params(cfg, dtype=torch.bfloat16, device="cuda:0"), strict=True)
        if args.fp8 or args.fp8_weights_only:
            model.enable_fp8()
        if args.fp8_weights_only:   # the Fp8Weight copies exist, every Linear still runs on the bf16 kernels
            hops.gemm_fp8_supported = lambda *a: False
            _C.gemm_fp8_supported = lambda *a: False
        inp = torch_inputs(cfg, B, T, h, w, L_txt, dtype=torch.bfloat16, device="cuda:0")
        side = None
        if args.stream_hammer:
            side = torch.cuda.Stream()
            ha = torch.randn(2048, 2048, device="cuda").bfloat16()
            hsrc = torch.empty(1 << 26, dtype=torch.uint8, device="cuda")
            hdst = torch.empty_like(hsrc)
        res = dict(rank=rank, bad_runs=[], first_div={}, diffs=[])
        with torch.inference_mode():
            single = model(**inp).float().cpu()
            sp = seqpar.enable(model, mode=args.mode)
            first, first_chain = None, None
            for run in range(args.runs):
                if args.poison != "none":
                    gen_holder["gen"].manual_seed(1000 + run)
                    poison_workspaces(model, sp, args.poison, gen_holder["gen"], _C)
                    gen_holder["on"] = True
                if side is not None:
                    with torch.cuda.stream(side):
                        for _ in range(6):
                            ha @ ha
                            hdst.copy_(hsrc)
                hops.reset()
                hops.enabled = args.instrument or args.checkpoints
                if args.checkpoints:
                    counter[0] = 0
                out = model(**inp)
                hops.enabled = False
                gen_holder["on"] = False
                names, chain = hops.collect()
                out = out.float().cpu()
                if first is None:
                    first, first_chain, first_names = out, chain, names
                    res["n_chain"] = len(chain)
                    res["finite_first"] = bool(torch.isfinite(out).all())
                    continue
                same = torch.equal(out, first) or (torch.isnan(out) == torch.isnan(first)).all() and torch.equal(torch.nan_to_num(out), torch.nan_to_num(first))
                div = None
                if args.instrument or args.checkpoints:
                    if names != first_names:
                        div = dict(kind="launch sequence differs", n=(len(names), len(first_names)))
                    else:
                        # "pre" checksums of OUTPUT buffers hold whatever the previous forward left there: only "post" entries
                        # can start a divergence (an input that differs was some earlier call's output)
                        for j, (x, y) in enumerate(zip(chain, first_chain)):
                            if x != y and names[j][2] == "post":
                                label, argi, phase = names[j]
                                pre_bad = [names[jj][1] for jj in range(j) if names[jj][0] == label and names[jj][2] == "pre"
                                           and chain[jj] != first_chain[jj]]
                                div = dict(entry=j, call=label, arg=argi, pre_args_that_differed=pre_bad)
                                break
                if not same or div is not None:
                    res["bad_runs"].append(run)
                    if div is not None:
                        key = json.dumps(div, sort_keys=True)
                        res["first_div"][key] = res["first_div"].get(key, 0) + 1
                    if not same and len(res["diffs"]) < 6:
                        res["diffs"].append(dict(run=run, **describe_diff(first, out, args.world, L_txt)))
            seqpar.disable(model)
        res["rel_l2_vs_single"] = float((first - single).norm() / single.norm())
        q.put(res)
    except BaseException:
        import traceback

        q.put(dict(rank=rank, error=traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--name", default="hd72_eager_split")
    ap.add_argument("--geom", type=lambda s: tuple(int(v) for v in s.split(",")), default=(2, 4, 8, 8, 64))
    ap.add_argument("--depth", type=lambda s: tuple(int(v) for v in s.split(",")), default=None, help="override depth,depth_single")
    ap.add_argument("--mode", default="allgather")
    ap.add_argument("--runs", type=int, default=40)
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--fp8-weights-only", action="store_true")
    ap.add_argument("--hammer", default="", help="comma list of contender processes: matmul, copy, small")
    ap.add_argument("--stream-hammer", action="store_true", help="contender work on a second stream of each rank process")
    ap.add_argument("--instrument", action="store_true")
    ap.add_argument("--checkpoints", action="store_true", help="light instrumentation: checksum of the residual stream after every block")
    ap.add_argument("--env", default="", help="comma list of K=V set in the environment of every process (library A/B switches)")
    ap.add_argument("--poison", default="none", choices=["none", "nan", "rand"])
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    for kv in args.env.split(","):
        if kv:
            k_, v_ = kv.split("=")
            os.environ[k_] = v_
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    hams = [ctx.Process(target=hammer_proc, args=(k,), daemon=True) for k in args.hammer.split(",") if k]
    for p in hams:
        p.start()
    if hams:
        time.sleep(8)   # let the contenders import torch and reach their loops
    procs = [ctx.Process(target=worker, args=(r, args, port, q)) for r in range(args.world)]
    t0 = time.time()
    for p in procs:
        p.start()
    out = []
    for _ in range(args.world):
        out.append(q.get(timeout=1800))
    for p in procs:
        p.join(timeout=60)
    for p in hams:
        p.kill()
    out.sort(key=lambda r: r["rank"])
    summary = dict(args={k: v for k, v in vars(args).items()}, seconds=round(time.time() - t0, 1), ranks=out)
    line = json.dumps(summary)
    print(line, flush=True)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            f.write(line + "\n")
    bad = sum(len(r.get("bad_runs", [])) for r in out) + sum(1 for r in out if "error" in r)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
