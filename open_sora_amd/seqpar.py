"""Sequence parallelism of the denoise step over the GPUs of one node (one process per GPU, RCCL over xGMI).

What the reference does (mmdit_model_forward, /root/reference/opensora/models/mmdit/distributed.py:580-683):
every rank receives the full inputs, keeps a contiguous 1/P chunk of the joint [txt ; img] token axis
(:609-624, `L % P == 0` asserted at :604-608), runs every token-wise op on its chunk, exchanges K/V inside
attention (ring P2P of KV chunks + LSE merge, :223-313, or Ulysses all-to-all, :473-495) and all-gathers the
prediction at the end (:678-679).

MI355X design: the exchange is ONE all-gather of K and of V^T per block instead of P-1 ring
hops — xGMI is a full point-to-point mesh, every rank can pull from all 7 peers at once, and the local flash
kernel then runs once over the whole key axis (segment-addressed: key j lives in gathered segment j / (L/P)), so
there is no per-hop LSE merge and no P small launches.  Every exchange runs on a SECOND HIP stream owned by the transport
(`DistTransport`: an event recorded on the compute stream behind the producing kernels, the collective issued on the
communication stream behind that event, a completion event the attention launch waits on -- stream-level, the host never
blocks), so the Q projection (+ the MLP-up projection in single blocks) runs on the compute stream beside the exchange.
No reduce-scatter is needed for inference (it is the backward of the all-gather).
The transport is an object (`all_gather / all_to_all / all_reduce_max`, each returning a handle with `wait()`): production =
`DistTransport` over `torch.distributed` (`nccl` = RCCL over xGMI); tests/local_transport.py runs P ranks as threads of ONE
process on ONE GPU with device-copy "collectives" on per-rank communication streams (tests/test_gpu_overlap.py), which
exercises the same event protocol and the overlap itself where no multi-GPU node is available.

Head-parallel exchange ("ulysses", the reference's other mode, distributed.py:473-495): when the head count divides
by P the block can instead all-to-all q, k, v from "my tokens, all heads" to "all tokens, my H/P heads", run the flash
kernel on whole sequences of H/P heads and all-to-all the result back.  Per rank and block it moves
4 (P-1)/P local tensors instead of 2 (P-1) — 4x less at P = 8, 2x less at P = 4, the same at P = 2 — and xGMI's
bandwidth per link is what bounds this small-hidden-size model at P > 2, so "auto" picks it for P >= 4.
The received chunks stay in their [source rank][batch][token] order: keys are addressed as P segments (the kernel's
segment layout), queries as P*B batches that share B key sets (osk_attention_fwd_bf16's kv_batches), and the output
lands directly in the layout the return all-to-all sends — no re-layout pass on the receive side.

Equal chunks keep every collective a plain `all_gather_into_tensor` / `all_to_all_single`; the final prediction is gathered as
[P, B, L/P, C] (rank 0 also projects its text rows, which are dropped after the gather) instead of the
reference's var-len gather (:39-112).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist
from torch import Tensor

from . import mmdit

BF16 = torch.bfloat16


class _Done:
    def wait(self):
        return True


class _EventWork:
    """completion of an exchange issued on the communication stream: wait() makes the CURRENT stream wait for it"""

    def __init__(self, event, device):
        self.event, self.device = event, device

    def wait(self):
        torch.cuda.current_stream(self.device).wait_event(self.event)
        return True


class DistTransport:
    """Collectives of one process group (`nccl` = RCCL in production, `gloo` in the CPU tests).  For device tensors every call
    is issued on this transport's communication stream behind an event of the calling (compute) stream and returns a handle
    whose wait() orders the caller's stream behind the exchange: compute queued between the call and the wait() overlaps it.
    The buffers handed in are persistent (SeqPar._bufs), so no allocator stream bookkeeping is needed."""

    def __init__(self, group=None, overlap: bool = True):
        self.group = group if group is not None else dist.group.WORLD
        self.P = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        self.overlap = overlap
        self._comm = {}

    def comm_stream(self, device):
        st = self._comm.get(str(device))
        if st is None:
            st = self._comm[str(device)] = torch.cuda.Stream(device)
        return st

    def _run(self, t: Tensor, fn):
        if not (t.is_cuda and self.overlap):
            fn()
            return _Done()
        cur, comm = torch.cuda.current_stream(t.device), self.comm_stream(t.device)
        ready = torch.cuda.Event()
        ready.record(cur)                    # everything queued so far (the kernels that produced the send buffer)
        with torch.cuda.stream(comm):
            comm.wait_event(ready)
            fn()                             # a synchronous collective orders `comm` behind the backend's own stream
            done = torch.cuda.Event()
            done.record(comm)
        return _EventWork(done, t.device)

    def all_gather(self, out: Tensor, inp: Tensor):
        """out (flat, P chunks) <- every rank's inp (flat, one chunk; may alias out's own chunk)"""
        return self._run(out, lambda: dist.all_gather_into_tensor(out, inp, group=self.group))

    def all_to_all(self, out: Tensor, inp: Tensor):
        """chunk s of out <- chunk `rank` of rank s's inp"""
        return self._run(out, lambda: dist.all_to_all_single(out, inp, group=self.group))

    def all_reduce_max(self, t: Tensor):
        """max over the ranks, in place (a few floats: the e4m3 scales of V in fp8 mode).  Issued on the communication stream like
        every other exchange (round 5 ran it as a blocking collective on the compute stream): what the caller queues before wait()
        -- the K copy into the gather buffer -- overlaps its latency."""
        return self._run(t, lambda: dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group))


class SeqPar:
    """Per-model sequence-parallel state: process group + the gathered K / V^T buffers (allocated once per
    geometry, reused by all 28/57 blocks: 2 * B * L * D * 2 bytes)."""

    def __init__(self, group=None, mode: str | None = None, transport=None):
        self.tp = transport if transport is not None else DistTransport(group)
        self.P, self.rank = self.tp.P, self.tp.rank
        self._bufs = {}
        self.exposed = None   # a list while bench.py measures exposed communication (see _timed_wait)
        self.mode = mode or os.environ.get("OSK_SP_MODE", "auto")   # "allgather" | "ulysses" | "auto"
        if self.mode not in ("allgather", "ulysses", "auto"):
            raise ValueError(f"unknown sequence-parallel mode {self.mode!r}")

    # ------------------------------------------------------------------ exposed-communication accounting (bench.py --gpus N)
    def _timed_wait(self, work, what: str):
        """wait() on an exchange; with `self.exposed` set to a list, bracket the wait with two events on the compute stream: the
        time between them is what the compute stream STALLED for this exchange -- its exposed part (0 when the exchange finished
        behind the kernels queued in front of the wait).  Events only; the host does not block."""
        rec = self.exposed
        if rec is None or not torch.cuda.is_available():
            work.wait()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        work.wait()
        e1.record()
        rec.append((what, e0, e1))

    def exposed_summary(self) -> dict:
        """ms the compute stream stalled per exchange kind since `exposed` was set (call after a synchronize)"""
        out: dict = {}
        for what, e0, e1 in self.exposed or []:
            out[what] = out.get(what, 0.0) + e0.elapsed_time(e1)
        return {k: round(v, 3) for k, v in out.items()}

    def head_parallel(self, H: int) -> bool:
        """exchange heads (all-to-all) instead of gathering K / V^T?"""
        if self.mode == "ulysses":
            assert H % self.P == 0, f"Expected {H} % {self.P} == 0"   # distributed.py:477-479
            return True
        return self.mode == "auto" and H % self.P == 0 and self.P >= 4

    # ------------------------------------------------------------------ sharding
    def shard_range(self, L: int, L_txt: int):
        """Token range [lo, hi) of the joint sequence owned by this rank, or None when sharding must be
        skipped because some rank would hold no image tokens (the reference then disables SP for the call,
        distributed.py:617-619)."""
        if self.P == 1:
            return None
        if L % self.P != 0:
            raise AssertionError(f"Expected {L} % {self.P} == 0")  # distributed.py:604-608
        Lloc = L // self.P
        if Lloc <= L_txt:  # rank 0 would be text-only (`0 in img_splits`)
            return None
        return self.rank * Lloc, (self.rank + 1) * Lloc

    # ------------------------------------------------------------------ K/V exchange
    MAX_GEOMETRIES = 4   # exchange-buffer sets kept (least recently used first out)

    def _cached(self, key, make):
        """exchange buffers of one (kind, geometry, device, stream), least-recently-used eviction: an i2v run that alternates between
        a few shapes keeps all of them (round 5 cleared the WHOLE cache when a fourth geometry appeared and re-allocated
        2 B L D buffers mid-loop); the set in use -- the one a captured hipGraph replays into -- is always the newest entry."""
        b = self._bufs.pop(key, None)
        if b is None:
            while len(self._bufs) >= self.MAX_GEOMETRIES:
                self._bufs.pop(next(iter(self._bufs)))
            b = make()
        self._bufs[key] = b
        return b

    def _buffers(self, B: int, Lloc: int, H: int, hd: int, device):
        def make():
            Lp = (Lloc + 63) // 64 * 64
            return (torch.empty(self.P, B, Lloc, H * hd, dtype=BF16, device=device),
                    torch.zeros(self.P, B, H, hd, Lp, dtype=BF16, device=device))

        return self._cached((B, Lloc, H, hd, str(device), mmdit._stream_key(device)), make)

    def _buffers8(self, B: int, Lloc: int, H: int, hd: int, device):
        def make():
            Lp = (Lloc + 63) // 64 * 64
            return (torch.empty(self.P, B, Lloc, H * hd, dtype=BF16, device=device),
                    torch.zeros(self.P, B, H, mmdit.ops().vt8_rows(hd), Lp, dtype=torch.uint8, device=device))

        return self._cached(("pv8", B, Lloc, H, hd, str(device), mmdit._stream_key(device)), make)

    def local_vt(self, B: int, Lloc: int, H: int, hd: int, device):
        """this rank's slot [B, H, hd, round_up(L/P, 64)] of the gathered V^T buffer (all-gather mode, bf16), or None: the projection
        may write V straight into it as V^T (osk_gemm_group_bf16's V^T task, round 6) -- gather_kv_start(..., vt_ready=True) then
        skips the osk_v_transpose_bf16 pass"""
        if self.head_parallel(H):
            return None
        return self._buffers(B, Lloc, H, hd, device)[1][self.rank]

    def gather_kv_start(self, ws, k: Tensor, v: Tensor, H: int, hd: int, pv8: bool = False, vt_ready: bool = False):
        """k, v: this rank's [B, L/P, D] views (K already normed + rotated with GLOBAL positions).  Starts the
        exchange of K and V and returns the handles; the caller keeps computing.
        pv8 (fp8 mode): V travels as e4m3 V^T (half the bytes); its per-(batch, head) scale must be the same on
        every rank -- the kernel accumulates P.V across all key segments -- so the local absmax is max-reduced first."""
        pv8 = pv8 and hd in (72, 128)
        if self.head_parallel(H):
            return self._heads_kv_start(k, v, H, hd, pv8)
        B, Lloc, _ = k.shape
        if pv8:
            sv = mmdit.v_scale_fp8(v, H, hd)
            # (co-residency rule of INTEGRATION.md section 4: the only REDUCING collective of this file; until its wait() below the
            #  compute stream runs the K copy only -- no hand-scheduled MFMA kernel overlaps RCCL's reduction kernel)
            wmax = self.tp.all_reduce_max(sv)                         # on the communication stream ...
            k_all, vt8_all = self._buffers8(B, Lloc, H, hd, k.device)
            mmdit.ops().copy_rows(k, k_all[self.rank])
            wk = self.tp.all_gather(k_all.view(-1), k_all[self.rank].view(-1))   # ... with the K copy and K's gather behind it
            self._timed_wait(wmax, "vmax")
            mmdit.ops().v_transpose_fp8(v, sv, vt8_all[self.rank], H, hd)
            wv = self.tp.all_gather(vt8_all.view(-1), vt8_all[self.rank].view(-1))
            return "pv8", k_all, vt8_all, sv, wk, wv
        k_all, vt_all = self._buffers(B, Lloc, H, hd, k.device)
        mmdit.ops().copy_rows(k, k_all[self.rank])
        if not vt_ready:                       # (vt_ready: the V projection already wrote local_vt())
            mmdit.ops().v_transpose(v, vt_all[self.rank], H, hd)
        wk = self.tp.all_gather(k_all.view(-1), k_all[self.rank].view(-1))
        wv = self.tp.all_gather(vt_all.view(-1), vt_all[self.rank].view(-1))
        return k_all, vt_all, wk, wv

    def attention(self, ws, pending, q: Tensor, out: Tensor, H: int, hd: int, score_bound: float = 0.0):
        """Local queries against the gathered keys: one launch, P key segments of L/P keys.  score_bound: the block's bound on
        |q . k| (mmdit._score_bound; it holds for every key of the joint sequence whichever rank projected it) -- the kernel's
        FAST body takes segmented / ragged key layouts since round 4, so sequence-parallel calls run it too."""
        if isinstance(pending[0], str) and pending[0] == "heads":
            return self._heads_attention(pending, q, out, H, hd, score_bound)
        B, Lloc, D = q.shape
        ops = mmdit.ops()
        if isinstance(pending[0], str):   # "pv8"
            _, k_all, vt8_all, sv, wk, wv = pending
            self._timed_wait(wk, "k")
            self._timed_wait(wv, "v")
            ops.attention_fwd_pv8(q, k_all[0], vt8_all, sv, out, H, hd, hd ** -0.5, n_seg=self.P, seg_len=Lloc,
                                  k_seg_stride=k_all.stride(0), vt_seg_stride=vt8_all.stride(0), q_prescaled=True,
                                  workspace=ops.attention_workspace(q.device))
            return
        k_all, vt_all, wk, wv = pending
        self._timed_wait(wk, "k")
        self._timed_wait(wv, "v")
        ops.attention_fwd(q, k_all[0], vt_all, out, H, hd, hd ** -0.5, n_seg=self.P, seg_len=Lloc,
                          k_seg_stride=k_all.stride(0), vt_seg_stride=vt_all.stride(0), q_prescaled=True,
                          workspace=ops.attention_workspace(q.device), score_bound=score_bound)

    # ------------------------------------------------------------------ head-parallel exchange (all-to-all)
    def _heads_buffers(self, B: int, Lloc: int, H: int, hd: int, device):
        def make():
            Hg, Lp = H // self.P, (Lloc + 63) // 64 * 64
            mk = lambda: torch.empty(self.P, B, Lloc, Hg * hd, dtype=BF16, device=device)
            b = dict(ks=mk(), kr=mk(), vs=mk(), vr=mk(), qs=mk(), qr=mk(), os=mk(), orr=mk(),
                     vt=torch.zeros(self.P, B, Hg, hd, Lp, dtype=BF16, device=device))
            if hd in (72, 128):   # fp8 mode: e4m3 V^T of the received chunks
                b["vt8"] = torch.zeros(self.P, B, Hg, mmdit.ops().vt8_rows(hd), Lp, dtype=torch.uint8, device=device)
            return b

        return self._cached(("heads", B, Lloc, H, hd, str(device), mmdit._stream_key(device)), make)

    def _to_head_chunks(self, dst: Tensor, x: Tensor):
        """[B, L/P, H*hd] (all heads of my tokens) -> dst [P, B, L/P, (H/P)*hd]: chunk j = head group j, for rank j.
        ONE osk_copy_rows_bf16 launch (chunk j of the source = columns j Dg ..): round 4 ran a torch permute-copy here, four per block."""
        B, Lloc, D = x.shape
        mmdit.ops().copy_rows(x.view(B, Lloc, self.P, D // self.P).permute(2, 0, 1, 3), dst)

    def _from_head_chunks(self, out: Tensor, src: Tensor):
        """the way back: src [P, B, L/P, Dg] (head group j of my tokens, from rank j) -> out [B, L/P, P x Dg], head groups side by side"""
        B, Lloc, D = out.shape
        mmdit.ops().copy_rows(src, out.view(B, Lloc, self.P, D // self.P).permute(2, 0, 1, 3))

    def _heads_kv_start(self, k: Tensor, v: Tensor, H: int, hd: int, pv8: bool = False):
        B, Lloc, _ = k.shape
        bufs = self._heads_buffers(B, Lloc, H, hd, k.device)
        self._to_head_chunks(bufs["ks"], k)
        wk = self.tp.all_to_all(bufs["kr"].view(-1), bufs["ks"].view(-1))
        self._to_head_chunks(bufs["vs"], v)
        wv = self.tp.all_to_all(bufs["vr"].view(-1), bufs["vs"].view(-1))
        return "heads", bufs, wk, wv, pv8

    def q_exchange_start(self, pending, q: Tensor, H: int, hd: int):
        """Head-parallel mode: start the all-to-all of the (normed, rotated, pre-scaled) queries and return the extended handle
        tuple -- whatever the caller queues before attention() overlaps it (single blocks: the MLP-up projection, the largest GEMM of
        the block; round 4 started this exchange inside attention() and waited on the spot).  All-gather mode: nothing to do."""
        if not (isinstance(pending[0], str) and pending[0] == "heads") or len(pending) > 5:
            return pending
        bufs = pending[1]
        self._to_head_chunks(bufs["qs"], q)
        wq = self.tp.all_to_all(bufs["qr"].view(-1), bufs["qs"].view(-1))
        return (*pending, wq)

    def _heads_attention(self, pending, q: Tensor, out: Tensor, H: int, hd: int, score_bound: float = 0.0):
        pending = self.q_exchange_start(pending, q, H, hd)     # (a caller that did not start it earlier)
        _, bufs, wk, wv, pv8, wq = pending
        B, Lloc, D = q.shape
        P, Hg = self.P, H // self.P
        self._timed_wait(wq, "q")
        self._timed_wait(wk, "k")
        self._timed_wait(wv, "v")
        # received chunk s = source rank s's tokens = key segment s; [P, B] is also the query "batch" axis
        ops = mmdit.ops()
        qr, kr, os_ = bufs["qr"].view(P * B, Lloc, Hg * hd), bufs["kr"], bufs["os"].view(P * B, Lloc, Hg * hd)
        if pv8:   # this rank holds the WHOLE sequence of its heads: the e4m3 scale needs no collective
            vr = bufs["vr"]
            sv = ops.v_scale_fp8(vr.view(P * B, Lloc, Hg * hd), Hg, hd).view(P, B, Hg).amax(0).contiguous()   # [B, Hg]
            vt8 = bufs["vt8"]
            ops.v_transpose_fp8(vr.view(P * B, Lloc, Hg * hd), sv.repeat(P, 1).contiguous(),
                                vt8.view(P * B, Hg, vt8.shape[-2], vt8.shape[-1]), Hg, hd)
            ops.attention_fwd_pv8(qr, kr[0], vt8, sv, os_, Hg, hd, hd ** -0.5, n_seg=P, seg_len=Lloc,
                                  k_seg_stride=kr.stride(0), vt_seg_stride=vt8.stride(0), q_prescaled=True, kv_batches=B,
                                  workspace=ops.attention_workspace(q.device))
        else:
            vt = bufs["vt"]
            ops.v_transpose(bufs["vr"].view(P * B, Lloc, Hg * hd), vt.view(P * B, Hg, hd, vt.shape[-1]), Hg, hd)
            ops.attention_fwd(qr, kr[0], vt, os_, Hg, hd, hd ** -0.5, n_seg=P, seg_len=Lloc, k_seg_stride=kr.stride(0),
                              vt_seg_stride=vt.stride(0), q_prescaled=True, kv_batches=B,
                              workspace=ops.attention_workspace(q.device), score_bound=score_bound)
        # chunk s of the output belongs to rank s's tokens: straight back, then head groups side by side
        self._timed_wait(self.tp.all_to_all(bufs["orr"].view(-1), bufs["os"].view(-1)), "o")
        self._from_head_chunks(out, bufs["orr"])

    # ------------------------------------------------------------------ output
    def gather_output(self, ws, project, C_out: int, L_txt: int) -> Tensor:
        """project(dst) writes this rank's [B, L/P, C_out] rows; returns the full image prediction
        [B, L_img, C_out] on every rank (gather_forward_split_backward_var_len, distributed.py:678-679)."""
        B, Lloc = ws.B, ws.L
        full = torch.empty(self.P, B, Lloc, C_out, dtype=BF16, device=ws.x.device)
        project(full[self.rank])
        self._timed_wait(self.tp.all_gather(full.view(-1), full[self.rank].view(-1)), "out")
        return full.permute(1, 0, 2, 3).reshape(B, self.P * Lloc, C_out)[:, L_txt:].contiguous()


def enable(model, group=None, mode: str | None = None, transport=None) -> SeqPar:
    """Shard `model`'s denoise step over `group` (default: WORLD).  The counterpart of installing
    MMDiTPolicy / Distributed*Processor through booster.boost (distributed.py:686-760): weights stay replicated,
    MMDiTModel.forward keeps its signature and returns the full prediction on every rank."""
    sp = SeqPar(group, mode, transport)
    model._sp = sp if sp.P > 1 else None
    return sp


def disable(model) -> None:
    model._sp = None
