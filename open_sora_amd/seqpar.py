"""Sequence parallelism of the denoise step over the GPUs of one node (one process per GPU, RCCL over xGMI).

What the reference does (mmdit_model_forward, /root/reference/opensora/models/mmdit/distributed.py:580-683):
every rank receives the full inputs, keeps a contiguous 1/P chunk of the joint [txt ; img] token axis
(:609-624, `L % P == 0` asserted at :604-608), runs every token-wise op on its chunk, exchanges K/V inside
attention (ring P2P of KV chunks + LSE merge, :223-313, or Ulysses all-to-all, :473-495) and all-gathers the
prediction at the end (:678-679).

MI355X design: the exchange is ONE all-gather of K and of V^T per block instead of P-1 ring
hops — xGMI is a full point-to-point mesh, every rank can pull from all 7 peers at once, and the local flash
kernel then runs once over the whole key axis (segment-addressed: key j lives in gathered segment j / (L/P)), so
there is no per-hop LSE merge and no P small launches.  The collectives are issued asynchronously
(ProcessGroupNCCL runs them on its own stream, ordered after the producing kernels) right after the K/V
projection + K-norm, and the Q projection (+ the MLP-up projection in single blocks) runs on the compute stream
meanwhile; the attention launch waits on the collectives' events (stream-level, the host never blocks).
No reduce-scatter is needed for inference (it is the backward of the all-gather).

Equal chunks keep every collective a plain `all_gather_into_tensor`; the final prediction is gathered as
[P, B, L/P, C] (rank 0 also projects its text rows, which are dropped after the gather) instead of the
reference's var-len gather (:39-112).
"""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import Tensor

from . import mmdit

BF16 = torch.bfloat16


class SeqPar:
    """Per-model sequence-parallel state: process group + the gathered K / V^T buffers (allocated once per
    geometry, reused by all 28/57 blocks: 2 * B * L * D * 2 bytes)."""

    def __init__(self, group=None):
        self.group = group if group is not None else dist.group.WORLD
        self.P = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        self._bufs = {}

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
    def _buffers(self, B: int, Lloc: int, H: int, hd: int, device):
        key = (B, Lloc, H, hd, str(device))
        b = self._bufs.get(key)
        if b is None:
            if len(self._bufs) > 2:
                self._bufs.clear()
            Lp = (Lloc + 63) // 64 * 64
            k_all = torch.empty(self.P, B, Lloc, H * hd, dtype=BF16, device=device)
            vt_all = torch.zeros(self.P, B, H, hd, Lp, dtype=BF16, device=device)
            b = self._bufs[key] = (k_all, vt_all)
        return b

    def gather_kv_start(self, ws, k: Tensor, v: Tensor, H: int, hd: int):
        """k, v: this rank's [B, L/P, D] views (K already normed + rotated with GLOBAL positions).  Starts the
        two all-gathers and returns the handles; the caller keeps computing."""
        B, Lloc, _ = k.shape
        k_all, vt_all = self._buffers(B, Lloc, H, hd, k.device)
        k_all[self.rank].copy_(k)
        mmdit.ops().v_transpose(v, vt_all[self.rank], H, hd)
        wk = dist.all_gather_into_tensor(k_all.view(-1), k_all[self.rank].view(-1), group=self.group, async_op=True)
        wv = dist.all_gather_into_tensor(vt_all.view(-1), vt_all[self.rank].view(-1), group=self.group, async_op=True)
        return k_all, vt_all, wk, wv

    def attention(self, ws, pending, q: Tensor, out: Tensor, H: int, hd: int):
        """Local queries against the gathered keys: one launch, P key segments of L/P keys."""
        k_all, vt_all, wk, wv = pending
        wk.wait()
        wv.wait()
        B, Lloc, D = q.shape
        mmdit.ops().attention_fwd(q, k_all[0], vt_all, out, H, hd, hd ** -0.5, n_seg=self.P, seg_len=Lloc,
                                  k_seg_stride=k_all.stride(0), vt_seg_stride=vt_all.stride(0), q_prescaled=True)

    # ------------------------------------------------------------------ output
    def gather_output(self, ws, project, C_out: int, L_txt: int) -> Tensor:
        """project(dst) writes this rank's [B, L/P, C_out] rows; returns the full image prediction
        [B, L_img, C_out] on every rank (gather_forward_split_backward_var_len, distributed.py:678-679)."""
        B, Lloc = ws.B, ws.L
        full = torch.empty(self.P, B, Lloc, C_out, dtype=BF16, device=ws.x.device)
        project(full[self.rank])
        dist.all_gather_into_tensor(full.view(-1), full[self.rank].view(-1), group=self.group)
        return full.permute(1, 0, 2, 3).reshape(B, self.P * Lloc, C_out)[:, L_txt:].contiguous()


def enable(model, group=None) -> SeqPar:
    """Shard `model`'s denoise step over `group` (default: WORLD).  The counterpart of installing
    MMDiTPolicy / Distributed*Processor through booster.boost (distributed.py:686-760): weights stay replicated,
    MMDiTModel.forward keeps its signature and returns the full prediction on every rank."""
    sp = SeqPar(group)
    model._sp = sp if sp.P > 1 else None
    return sp


def disable(model) -> None:
    model._sp = None
