"""TEST INFRASTRUCTURE ONLY — CPU restatement (torch, any float dtype) of the reference's rectified-flow sampling
loop around the denoiser, and of the api_fn pipeline either side of it.

Not the product and never imported by it: only tests/ use this file, as the checker.  The arithmetic runs in the dtype
of the tensors handed in: fp32 = truth, bf16 = the reference-precision comparator (the reference keeps the latent
state and does the update in the model dtype, bf16).  Pinned against the reference's own source text (executed through
oracle.ref_loader.extract_defs) in tests/test_sampling_oracle.py where /root/reference is mounted.

Paths below are relative to /root/reference/opensora/utils.
"""
from __future__ import annotations

import math

import torch
from torch import Tensor

from . import mmdit_oracle as O


def oscillation_gs(scale: float, i: int, force_num: int = 10) -> float:
    """get_oscillation_gs (sampling.py:120-133): the full scale during the first force_num steps and on even steps."""
    return scale if (i < force_num or i % 2 == 0) else 1.0


def schedule(num_steps: int, image_seq_len: int, num_frames: int, shift_alpha=None, shift: bool = True) -> list:
    """get_schedule + time_shift + get_res_lin_function (sampling.py:295-332): t_i = 1 - i/N in f32, then
    t -> a t / (1 + (a - 1) t) with a = (1 + 2 (L - 256) / 3840) sqrt(T) unless given."""
    ts = torch.linspace(1, 0, num_steps + 1)
    if shift:
        if shift_alpha is None:
            shift_alpha = (1.0 + (3.0 - 1.0) / (4096 - 256) * (image_seq_len - 256)) * math.sqrt(num_frames)
        ts = shift_alpha * ts / (1 + (shift_alpha - 1) * ts)
    return ts.tolist()


def pack(x: Tensor, p: int = 2) -> Tensor:
    """'b c t (h ph) (w pw) -> b (t h w) (c ph pw)' (sampling.py:375-378)"""
    b, c, t, H, W = x.shape
    return x.reshape(b, c, t, H // p, p, W // p, p).permute(0, 2, 3, 5, 1, 4, 6).reshape(b, t * (H // p) * (W // p), c * p * p)


def unpack(x: Tensor, h: int, w: int, t: int, p: int = 2) -> Tensor:
    """'b (t h w) (c ph pw) -> b c t (h ph) (w pw)' with h, w in patch units (sampling.py:381-393)"""
    b, _, cpp = x.shape
    c = cpp // (p * p)
    return x.reshape(b, t, h, w, c, p, p).permute(0, 4, 1, 2, 5, 3, 6).reshape(b, c, t, h * p, w * p)


def grid_ids(bs: int, t: int, hp: int, wp: int, n_txt: int, dtype):
    """img_ids = (t, h, w) patch grid, txt_ids = 0 (sampling.py:437-447)"""
    ids = torch.zeros(t, hp, wp, 3)
    ids[..., 0] += torch.arange(t)[:, None, None]
    ids[..., 1] += torch.arange(hp)[None, :, None]
    ids[..., 2] += torch.arange(wp)[None, None, :]
    return ids.reshape(1, -1, 3).repeat(bs, 1, 1).to(dtype), torch.zeros(bs, n_txt, 3, dtype=dtype)


def i2v_denoise(model_fn, img3: Tensor, timesteps, guidance: float, guidance_img: float, masks: Tensor,
                masked_ref: Tensor, text_osci=False, image_osci=False, scale_temporal_osci=False, patch_size: int = 2,
                **model_kwargs) -> Tensor:
    """I2VDenoiser.denoise (sampling.py:158-226).  img3 = the latent tokens tripled (cond | uncond | uncond_2); the
    state and every elementwise op are in img3.dtype.  model_fn(img=, cond=, timesteps=, guidance=, **model_kwargs)."""
    n3 = img3.shape[0]
    n = n3 // 3
    dt = img3.dtype
    b, c, t, d3, d4 = masked_ref.shape   # the reference names the last two (w, h) and repeats (h, w): square or not, see below
    cond = pack(torch.cat((masks, masked_ref), 1), patch_size)
    cond3 = torch.cat([cond, cond, torch.zeros_like(cond)], 0)          # :187-189: branch 3 drops the image condition
    g_vec = torch.full((n3,), guidance, dtype=dt)
    x = img3[:n]
    for i, (t_curr, t_prev) in enumerate(zip(timesteps[:-1], timesteps[1:])):
        t_vec = torch.full((n3,), t_curr, dtype=dt)                     # :183-185: rounded to the model dtype
        pred = model_fn(img=torch.cat([x, x, x], 0), cond=cond3, timesteps=t_vec, guidance=g_vec, **model_kwargs)
        tg = oscillation_gs(guidance, i) if text_osci else guidance
        ig = oscillation_gs(guidance_img, i) if image_osci else guidance_img
        pc, pu, pu2 = pred.chunk(3, 0)
        if ig > 1.0 and scale_temporal_osci:                            # :205-213
            upper = torch.linspace(ig, 1.0, len(timesteps))[i]
            ramp = torch.linspace(1.0, float(upper), t)[None, None, :, None, None].repeat(b, c, 1, d4, d3)
            ig = pack(ramp, patch_size).to(dt)
        v = pu2 + ig * (pu - pu2) + tg * (pc - pu)                      # :216
        x = x + (t_prev - t_curr) * v                                   # :219
    return x


def mmdit_fn(sd: dict, cfg: dict):
    """the denoiser as the loop calls it: the oracle forward on a state dict"""
    def fn(**kw):
        kw = {k: v for k, v in kw.items() if v is not None}
        if not cfg.get("guidance_embed", False):
            kw.pop("guidance", None)
        return O.forward(sd, cfg, **kw)
    return fn
