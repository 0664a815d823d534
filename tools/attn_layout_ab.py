#!/usr/bin/env python
"""Same-process A/B of the two work-unit layouts of the bounded head_dim-72 attention (256-row units / the wide 512-row units), with and
without the tail-split workspace, at the launch shapes of CFG batch 1 and of sequence-parallel ranks -- where csrc/attention_params.h::
attn_wide_path has to choose by estimated rounds of the chip.  One JSON line per shape."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_sora_amd import _C

dev = torch.device("cuda")
H, hd, L = 16, 72, 16896
# (query batches, heads per launch, Lq, key segments, segment length, key batches, what)
SHAPES = [(3, 16, L, 1, L, 0, "B=3 single GPU"), (1, 16, L, 1, L, 0, "B=1 single GPU"),
          (3, 16, L // 2, 2, L // 2, 0, "P=2 rank, K / V^T all-gather"), (1, 16, L // 2, 2, L // 2, 0, "B=1 P=2 rank, all-gather"),
          (12, 4, L // 4, 4, L // 4, 3, "P=4 rank, head exchange"), (24, 2, L // 8, 8, L // 8, 3, "P=8 rank, head exchange"),
          (4, 4, L // 4, 4, L // 4, 1, "B=1 P=4 rank, head exchange"), (3, 16, L // 4, 4, L // 4, 0, "P=4 rank, all-gather")]
ws = _C.attention_workspace(dev)
for Bq, Hh, Lq, n_seg, seg, kvb, what in SHAPES:
    D = Hh * hd
    nb = kvb or Bq
    q = torch.randn(Bq, Lq, D, device=dev).to(torch.bfloat16)
    k = torch.randn(n_seg, nb, seg, D, device=dev).to(torch.bfloat16)
    v = torch.randn(n_seg, nb, seg, D, device=dev).to(torch.bfloat16)
    segp = (seg + 63) // 64 * 64
    vt = torch.zeros(n_seg, nb, Hh, hd, segp, dtype=torch.bfloat16, device=dev)
    _C.v_transpose(v.view(n_seg * nb, seg, D), vt.view(n_seg * nb, Hh, hd, segp), Hh, hd)
    out = torch.empty(Bq, Lq, D, dtype=torch.bfloat16, device=dev)
    qn = q.float().view(Bq, Lq, Hh, hd).norm(dim=-1).amax().item() * hd ** -0.5 * 1.4426950408889634
    kn = k.float().view(-1, Hh, hd).norm(dim=-1).amax().item()
    bound = min(qn * kn, 55.0)
    kw = dict(n_seg=n_seg, seg_len=seg, k_seg_stride=k.stride(0), vt_seg_stride=vt.stride(0), kv_batches=kvb, score_bound=bound)
    rec = {"what": what, "shape": [Bq, Hh, Lq, n_seg, seg], "ms": {}}
    rec["estimate"] = list(_C.attention_launch_shape(Bq, Hh, Lq, n_seg, seg, hd, bound, ws.numel()))
    for rows in (256, 512):
        _C.lib.osk_attention_rows_override(rows)
        for w, tag in ((ws, "split"), (None, "nosplit")):
            for _ in range(2):
                _C.attention_fwd(q, k[0], vt, out, Hh, hd, hd ** -0.5, workspace=w, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                _C.attention_fwd(q, k[0], vt, out, Hh, hd, hd ** -0.5, workspace=w, **kw)
            e1.record()
            torch.cuda.synchronize()
            rec["ms"][f"{rows}_{tag}"] = round(e0.elapsed_time(e1) / 6, 4)
    _C.lib.osk_attention_rows_override(0)
    rec["best"] = min(rec["ms"], key=rec["ms"].get)
    est = f"{rec['estimate'][1]}_split"
    rec["loss_of_estimate"] = round(rec["ms"][est] / rec["ms"][rec["best"]] - 1, 4)
    print(json.dumps(rec), flush=True)
