"""TEST INFRASTRUCTURE ONLY -- P sequence-parallel ranks as THREADS of one process on ONE GPU.

open_sora_amd.seqpar talks to its peers through a transport object (`all_gather / all_to_all / all_reduce_max`, each returning
a handle with wait()).  Production binds it to torch.distributed (RCCL); here the "collectives" are device copies that every
rank issues on its own communication stream, ordered by the same event protocol DistTransport uses:

    caller's compute stream --record--> ready[r]          (my send buffer is final behind this event)
    rank r's comm stream: wait ready[s] of every peer s, pull peer s's chunk, --record--> done[r]
    handle.wait(): the caller's compute stream waits for done[*] of ALL ranks -- my data has arrived AND every peer has read my
                   send buffer (what completion of an RCCL collective guarantees before the buffer may be rewritten)

Host-side the ranks meet at thread barriers only to exchange tensor handles and events; the GPU work of all ranks (P compute
streams + P communication streams) runs concurrently on the one device, which is the co-residency the production overlap
creates (exchange kernels beside the Q / MLP-up GEMMs).  `overlap=False` issues the copies on the compute stream instead:
the serial order the overlapped run must reproduce bit for bit."""
from __future__ import annotations

import threading

import torch


class LocalWorld:
    def __init__(self, P: int):
        self.P = P
        self.barrier = threading.Barrier(P)
        self.post = [None] * P      # (tensor, ready event) posted by each rank for the collective in flight
        self.done = [None] * P


class _Work:
    def __init__(self, events, device):
        self.events, self.device = events, device

    def wait(self):
        if self.events:
            cur = torch.cuda.current_stream(self.device)
            for e in self.events:
                cur.wait_event(e)
        return True


class LocalTransport:
    def __init__(self, world: LocalWorld, rank: int, device, overlap: bool = True):
        self.w, self.P, self.rank, self.device, self.overlap = world, world.P, rank, torch.device(device), overlap
        self.cuda = self.device.type == "cuda"   # "cpu": the same thread protocol without streams (host-logic test)
        self.comm = torch.cuda.Stream(self.device) if self.cuda else None
        self.calls = 0

    def _exchange(self, inp: torch.Tensor, pull):
        """post my send tensor, meet the peers, run pull(posts) on my communication stream, publish my completion event"""
        w = self.w
        if not self.cuda:
            w.post[self.rank] = (inp, None)
            w.barrier.wait()
            pull(list(w.post))
            w.barrier.wait()                               # every rank has read what it needed
            self.calls += 1
            return _Work([], self.device)
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        w.post[self.rank] = (inp, ready)
        w.barrier.wait()                                   # every rank has posted
        posts = list(w.post)
        stream = self.comm if self.overlap else cur
        with torch.cuda.stream(stream):
            for s, (_, ev) in enumerate(posts):
                if s != self.rank:
                    stream.wait_event(ev)
            stream.wait_event(ready)
            pull(posts)
            done = torch.cuda.Event()
            done.record(stream)
        w.done[self.rank] = done
        w.barrier.wait()                                   # every rank has enqueued its pulls
        events = list(w.done)
        w.barrier.wait()                                   # the slots may be reused
        self.calls += 1
        return _Work(events, self.device)

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        P = self.P
        o = out.view(P, -1)

        def pull(posts):
            for s, (src, _) in enumerate(posts):
                if o[s].data_ptr() != src.data_ptr():      # my own chunk usually IS the send buffer
                    o[s].copy_(src.view(-1))

        return self._exchange(inp, pull)

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor):
        P, r = self.P, self.rank
        o = out.view(P, -1)

        def pull(posts):
            for s, (src, _) in enumerate(posts):
                o[s].copy_(src.view(P, -1)[r])

        return self._exchange(inp, pull)

    def all_reduce_max(self, t: torch.Tensor):
        mine = t.clone()

        def pull(posts):
            for s, (src, _) in enumerate(posts):
                torch.maximum(t, src, out=t)

        self._exchange(mine, pull).wait()
        return _Work([], self.device)


def run_ranks(P: int, fn, device="cuda:0"):
    """fn(rank, transport) on P threads, each inside its own compute stream; returns the list of results (exceptions re-raised)"""
    world = LocalWorld(P)
    res, err = [None] * P, [None] * P

    def body(r):
        try:
            if torch.device(device).type == "cuda":
                torch.cuda.set_device(device)
                st = torch.cuda.Stream(device)
                with torch.cuda.stream(st):
                    res[r] = fn(r, world)
                st.synchronize()
            else:
                res[r] = fn(r, world)
        except BaseException as e:   # noqa: BLE001 -- surfaced below; a dead rank must not leave the others at a barrier
            err[r] = e
            world.barrier.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(P)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return res
