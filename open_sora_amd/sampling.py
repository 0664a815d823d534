"""Host side of the rectified-flow sampler around the denoiser (the caller of the hot path).

Mirrors, by name and argument meaning, the pieces of /root/reference/opensora/utils/sampling.py that sit on
either side of MMDiTModel.forward:
    get_oscillation_gs :120-133    time_shift / get_res_lin_function / get_schedule :295-332
    get_noise :335-372             pack / unpack :375-393         prepare_ids (from prepare :431-447)
    I2VDenoiser.denoise :159-226   (CFG triple, oscillating guidance, Euler update)
Schedules are Python floats; tensors stay on the device; the CFG combine + Euler update is one HIP kernel
(osk_cfg_euler_bf16) instead of six bf16 elementwise launches.
"""
from __future__ import annotations

import math
import os

import torch
from torch import Tensor

from .mmdit import ops as _ops  # kernel table (HIP library; tests may swap in their CPU emulation)


def get_oscillation_gs(guidance_scale: float, i: int, force_num: int = 10) -> float:
    """sampling.py:120-133: full guidance for the first `force_num` steps and on even steps, 1.0 otherwise."""
    return guidance_scale if (i < force_num or i % 2 == 0) else 1.0


def time_shift(alpha: float, t):
    return alpha * t / (1 + (alpha - 1) * t)


def get_res_lin_function(x1: float = 256, y1: float = 1, x2: float = 4096, y2: float = 3):
    m = (y2 - y1) / (x2 - x1)
    b = y1 - m * x1
    return lambda x: m * x + b


def get_schedule(num_steps: int, image_seq_len: int, num_frames: int, shift_alpha: float | None = None,
                 base_shift: float = 1, max_shift: float = 3, shift: bool = True) -> list[float]:
    """sampling.py:307-332: linspace(1, 0, N+1) in f32, optionally shifted by alpha(seq_len) * sqrt(frames)."""
    ts = torch.linspace(1, 0, num_steps + 1)
    if shift:
        if shift_alpha is None:
            shift_alpha = get_res_lin_function(y1=base_shift, y2=max_shift)(image_seq_len)
            shift_alpha *= math.sqrt(num_frames)
        ts = time_shift(shift_alpha, ts)
    return ts.tolist()


def _ae_compression() -> int:
    return int(os.environ.get("AE_SPATIAL_COMPRESSION", 16))


def get_noise(num_samples: int, height: int, width: int, num_frames: int, device, dtype, seed: int,
              patch_size: int = 2, channel: int = 16) -> Tensor:
    """sampling.py:335-372 (device generator seeded per call)."""
    D = _ae_compression()
    gen = torch.Generator(device=device).manual_seed(seed)
    return torch.randn(num_samples, channel, num_frames, patch_size * math.ceil(height / D),
                       patch_size * math.ceil(width / D), device=device, dtype=dtype, generator=gen)


def pack(x: Tensor, patch_size: int = 2) -> Tensor:
    """'b c t (h ph) (w pw) -> b (t h w) (c ph pw)' (sampling.py:375-378)."""
    b, c, t, H, W = x.shape
    p = patch_size
    x = x.reshape(b, c, t, H // p, p, W // p, p)
    return x.permute(0, 2, 3, 5, 1, 4, 6).reshape(b, t * (H // p) * (W // p), c * p * p)


def unpack(x: Tensor, height: int, width: int, num_frames: int, patch_size: int = 2) -> Tensor:
    """'b (t h w) (c ph pw) -> b c t (h ph) (w pw)' (sampling.py:381-393)."""
    D = _ae_compression()
    h, w, p = math.ceil(height / D), math.ceil(width / D), patch_size
    b, _, cpp = x.shape
    c = cpp // (p * p)
    x = x.reshape(b, num_frames, h, w, c, p, p)
    return x.permute(0, 4, 1, 2, 5, 3, 6).reshape(b, c, num_frames, h * p, w * p)


def prepare_ids(bs: int, t: int, h: int, w: int, n_txt: int, device, dtype, patch_size: int = 2):
    """img_ids[t,h,w] = (t,h,w) in patch units, txt_ids = 0 (sampling.py:437-447,455-457)."""
    hp, wp = h // patch_size, w // patch_size
    ids = torch.zeros(t, hp, wp, 3)
    ids[..., 0] += torch.arange(t)[:, None, None]
    ids[..., 1] += torch.arange(hp)[None, :, None]
    ids[..., 2] += torch.arange(wp)[None, None, :]
    img_ids = ids.reshape(1, t * hp * wp, 3).repeat(bs, 1, 1).to(device, dtype)
    txt_ids = torch.zeros(bs, n_txt, 3, device=device, dtype=dtype)
    return img_ids, txt_ids


_SIDE_STREAMS: dict = {}   # device -> the capture stream of denoise(hip_graph=True)


class I2VDenoiser:
    """sampling.py:158-245.  `denoise(model, img=..., timesteps=[...], guidance=..., guidance_img=..., masks=...,
    masked_ref=..., img_ids=..., txt=..., txt_ids=..., y_vec=..., [text_osci, image_osci, scale_temporal_osci,
    patch_size, sigma_min])` with img/txt/... already tripled (cond | uncond | uncond_2).

    Not in the reference: `hip_graph=True` replays the model forward of steps 1.. from a hipGraph captured after step 0
    (every entry point of the library is capturable: no allocation, no synchronisation; the per-step inputs -- the
    latent triple and the timestep vector -- live in fixed buffers).  Same results bit for bit.  Measured on this host:
    no gain at either end -- the XL step's ~540 launches hide behind 150 ms of kernels, and even the S model's 2.6 ms
    forward replays in the same 2.6 ms (`bench.py`: cpu_baseline.cfg1.gpu_hipgraph_ms) -- so it is off by default and
    meant for hosts whose Python dispatch is slower than the GPU."""

    def denoise(self, model, **kwargs) -> Tensor:
        img = kwargs.pop("img")
        timesteps = kwargs.pop("timesteps")
        guidance = kwargs.pop("guidance")
        guidance_img = kwargs.pop("guidance_img")
        masks = kwargs.pop("masks")
        masked_ref = kwargs.pop("masked_ref")
        kwargs.pop("sigma_min", None)
        text_osci = kwargs.pop("text_osci", False)
        image_osci = kwargs.pop("image_osci", False)
        scale_temporal_osci = kwargs.pop("scale_temporal_osci", False)
        patch_size = kwargs.pop("patch_size", 2)
        hip_graph = bool(kwargs.pop("hip_graph", False))

        n3 = img.shape[0]
        n = n3 // 3
        dev, dt = img.device, img.dtype
        guidance_vec = torch.full((n3,), guidance, device=dev, dtype=dt)
        b, c, t, w_, h_ = masked_ref.size()
        cond = pack(torch.cat((masks, masked_ref), dim=1), patch_size=patch_size)
        cond3 = torch.cat([cond, cond, torch.zeros_like(cond)], dim=0)  # 3rd branch drops the image condition
        # private state (the ping-pong below must not write into the caller's tensor), kept in bf16 -- the dtype the
        # update kernel computes in and the reference samples in (configs/diffusion/inference/256px.py:4); a model in
        # another dtype is fed / read through casts at this boundary
        x = img[:n].to(torch.bfloat16, copy=True).contiguous()
        x_next = torch.empty_like(x)
        img3 = torch.empty(n3, *x.shape[1:], device=dev, dtype=dt)
        t_vec = torch.empty(n3, dtype=dt, device=dev)
        graph, pred = None, None
        if hip_graph:
            # capture needs a non-default stream, and the model's workspaces are keyed on the stream: run the WHOLE loop (the
            # eager step 0 as well) on one side stream, so capture re-uses step 0's workspaces instead of allocating a second
            # set from the graph's private pool that would stay pinned for the model's lifetime
            # ONE side stream per device for the lifetime of the process (the workspace key contains the stream handle: a fresh
            # stream per call would allocate a fresh activation workspace set per sampling run -- ADVICE r3)
            side = _SIDE_STREAMS.get(str(dev))
            if side is None:
                side = _SIDE_STREAMS[str(dev)] = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                out = self._loop(model, kwargs, x, x_next, img3, t_vec, cond3, guidance_vec, timesteps, guidance, guidance_img,
                                 text_osci, image_osci, scale_temporal_osci, patch_size, b, c, t, h_, w_, dev, True, side)
            torch.cuda.current_stream(dev).wait_stream(side)
            return out.to(dt)
        return self._loop(model, kwargs, x, x_next, img3, t_vec, cond3, guidance_vec, timesteps, guidance, guidance_img, text_osci,
                          image_osci, scale_temporal_osci, patch_size, b, c, t, h_, w_, dev, False, None).to(dt)

    @staticmethod
    def _loop(model, kwargs, x, x_next, img3, t_vec, cond3, guidance_vec, timesteps, guidance, guidance_img, text_osci, image_osci,
              scale_temporal_osci, patch_size, b, c, t, h_, w_, dev, hip_graph, side):
        graph, pred = None, None
        for i, (t_curr, t_prev) in enumerate(zip(timesteps[:-1], timesteps[1:])):
            t_vec.fill_(t_curr)
            if _ops().copy_rows_ok(x, img3[:x.shape[0]]):
                # the CFG triple's input = the latents three times over (sampling.py:196-201): one broadcast launch for one sample
                for j in ([None] if x.shape[0] == 1 else range(3)):
                    _ops().copy_rows(x, img3 if j is None else img3[j * x.shape[0]:(j + 1) * x.shape[0]])
            else:
                img3.view(3, *x.shape).copy_(x.unsqueeze(0).expand(3, *x.shape))
            if graph is not None:
                graph.replay()                    # writes the captured `pred`
            else:
                pred = model(img=img3, **kwargs, cond=cond3, timesteps=t_vec, guidance=guidance_vec)
                if hip_graph and i == 0 and len(timesteps) > 2:
                    # step 0 ran eagerly (it built the plans and workspaces); capture the same call for the rest
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=side):
                        pred_g = model(img=img3, **kwargs, cond=cond3, timesteps=t_vec, guidance=guidance_vec)
            text_gs = get_oscillation_gs(guidance, i) if text_osci else guidance
            image_gs = get_oscillation_gs(guidance_img, i) if image_osci else guidance_img
            gvec = None
            if image_gs > 1.0 and scale_temporal_osci:
                upper = torch.linspace(image_gs, 1.0, len(timesteps))[i]
                ramp = torch.linspace(1.0, float(upper), t)[None, None, :, None, None].repeat(b, c, 1, h_, w_)
                gvec = pack(ramp, patch_size=patch_size).to(dev, torch.float32).contiguous()
                image_gs = 1.0
            _ops().cfg_euler(pred.to(torch.bfloat16).contiguous(), x, x_next, float(text_gs), float(image_gs),
                             float(t_prev - t_curr), gvec)
            x, x_next = x_next, x
            if graph is not None:
                pred = pred_g                     # from now on the forward's output is the captured tensor
        return x

    def prepare_guidance(self, text: list, optional_models: dict, device, dtype, **kwargs):
        ret = {"guidance_img": kwargs.pop("guidance_img")}
        neg = kwargs.get("neg", None)
        if neg is None:
            neg = [""] * len(text)
        return text + neg + neg, ret
