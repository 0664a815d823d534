"""The inference API around the hot path: what `scripts/diffusion/inference.py` calls.

Mirrors, by name, arguments and behaviour:
    SamplingMethod                       opensora/utils/inference.py:16-18
    SamplingOption, sanitize_…           opensora/utils/sampling.py:27-117
    get_image_size (+ helpers)           opensora/datasets/aspect.py:4-58,66-76,133-139
    prepare                              opensora/utils/sampling.py:401-459   (text encoders are injected callables)
    prepare_inference_condition          opensora/utils/inference.py:283-351
    collect_references_batch             opensora/utils/inference.py:216-280  (media reading is injected: `reader`)
    DistilledDenoiser, SamplingMethodDict  opensora/utils/sampling.py:253-288
    prepare_api / api_fn                 opensora/utils/sampling.py:562-726
The denoiser (`MMDiTModel.forward`) and the VAE (`decode` / `encode`) behind it are the HIP modules of this package;
T5 / CLIP (`model_t5(prompt, added_tokens=, seq_align=)`, `model_clip(prompt)`) and the media reader are the
caller's: they are outside the hot path and are only called, never re-implemented, here.
"""
from __future__ import annotations

import math
import os
import random
from dataclasses import dataclass, replace
from enum import Enum

import torch
from torch import Tensor

from .sampling import I2VDenoiser, get_noise, get_schedule, pack, unpack


class SamplingMethod(Enum):
    I2V = "i2v"          # open sora video generation
    DISTILLED = "distill"  # flux image generation


@dataclass
class SamplingOption:
    """sampling.py:27-81 (same fields, same defaults)."""
    width: int | None = None
    height: int | None = None
    resolution: str | None = None
    aspect_ratio: str | None = None
    num_frames: int = 1
    num_steps: int = 50
    guidance: float = 4.0
    text_osci: bool = False
    guidance_img: float | None = None
    image_osci: bool = False
    scale_temporal_osci: bool = False
    seed: int | None = None
    shift: bool = True
    method: str | SamplingMethod = SamplingMethod.I2V
    temporal_reduction: int = 1
    is_causal_vae: bool = False
    flow_shift: float | None = None


# ------------------------------------------------------------------------------------------------ aspect.py
ASPECT_RATIO_LD_LIST = ["2.39:1", "2:1", "16:9", "1.85:1", "9:16", "5:8", "3:2", "4:3", "1:1"]  # width:height


def get_aspect_ratios_dict(total_pixels: int = 256 * 256, training: bool = True) -> dict[str, tuple[int, int]]:
    """aspect.py:22-58."""
    D = int(os.environ.get("AE_SPATIAL_COMPRESSION", 16))
    out, vertical = {}, {}
    for ratio in ASPECT_RATIO_LD_LIST:
        wr, hr = map(float, ratio.split(":"))
        width = int(math.sqrt(total_pixels * (wr / hr)) // D) * D
        height = int((total_pixels / width) // D) * D
        if training:  # adjust aspect ratio to match total pixels
            diff = abs(height * width - total_pixels)
            for h, w in [(height - D, width), (height + D, width), (height, width - D), (height, width + D)]:
                if abs(h * w - total_pixels) < diff:
                    height, width = h, w
                    diff = abs(h * w - total_pixels)
        if (height, width) not in out.values() or not training:
            out[ratio] = (height, width)
            vertical[":".join(ratio.split(":")[::-1])] = (width, height)
    out.update(vertical)
    return out


def get_num_pexels_from_name(resolution: str) -> int:
    """aspect.py:66-76: "256px" -> 256^2, "720p" -> 720^2 * 16 / 9."""
    resolution = resolution.split("_")[0]
    if resolution.endswith("px"):
        return int(resolution[:-2]) ** 2
    if resolution.endswith("p"):
        size = int(resolution[:-1])
        return int(size * size / 9 * 16)
    raise ValueError(f"Invalid resolution {resolution}")


def get_image_size(resolution: str, ar_ratio: str, training: bool = True) -> tuple[int, int]:
    """aspect.py:133-139."""
    ar_dict = get_aspect_ratios_dict(get_num_pexels_from_name(resolution), training)
    assert ar_ratio in ar_dict, f"Aspect ratio {ar_ratio} not found"
    return ar_dict[ar_ratio]


def sanitize_sampling_option(sampling_option: SamplingOption) -> SamplingOption:
    """sampling.py:84-117."""
    if sampling_option.resolution is not None or sampling_option.aspect_ratio is not None:
        assert sampling_option.resolution is not None and sampling_option.aspect_ratio is not None, \
            "Both resolution and aspect ratio must be provided"
        height, width = get_image_size(sampling_option.resolution, sampling_option.aspect_ratio, training=False)
    else:
        assert sampling_option.height is not None and sampling_option.width is not None, \
            "Both height and width must be provided"
        height, width = sampling_option.height, sampling_option.width
    height = (height // 16 + (1 if height % 16 else 0)) * 16
    width = (width // 16 + (1 if width % 16 else 0)) * 16
    rep = dict(height=height, width=width)
    if isinstance(sampling_option.method, str):
        rep["method"] = SamplingMethod(sampling_option.method)
    return replace(sampling_option, **rep)


# ------------------------------------------------------------------------------------------------ prepare
def prepare(t5, clip, img: Tensor, prompt: str | list[str], seq_align: int = 1, patch_size: int = 2) -> dict[str, Tensor]:
    """sampling.py:401-459: pack the latent, build (t,h,w) position ids, call the two text encoders."""
    bs, c, t, h, w = img.shape
    device, dtype = img.device, img.dtype
    if isinstance(prompt, str):
        prompt = [prompt]
    if bs != len(prompt):
        bs = len(prompt)
    img = pack(img, patch_size=patch_size)
    if img.shape[0] != bs:
        img = img.repeat(bs // img.shape[0], 1, 1)            # "b ... -> (repeat b) ..."
    hp, wp = h // patch_size, w // patch_size
    img_ids = torch.zeros(t, hp, wp, 3)
    img_ids[..., 0] = img_ids[..., 0] + torch.arange(t)[:, None, None]
    img_ids[..., 1] = img_ids[..., 1] + torch.arange(hp)[None, :, None]
    img_ids[..., 2] = img_ids[..., 2] + torch.arange(wp)[None, None, :]
    img_ids = img_ids.reshape(1, t * hp * wp, 3).repeat(bs, 1, 1)
    txt = t5(prompt, added_tokens=img_ids.shape[1], seq_align=seq_align)
    if txt.shape[0] == 1 and bs > 1:
        txt = txt.repeat(bs, *([1] * (txt.dim() - 1)))
    txt_ids = torch.zeros(bs, txt.shape[1], 3)
    vec = clip(prompt)
    if vec.shape[0] == 1 and bs > 1:
        vec = vec.repeat(bs, *([1] * (vec.dim() - 1)))
    return {"img": img, "img_ids": img_ids.to(device, dtype), "txt": txt.to(device, dtype),
            "txt_ids": txt_ids.to(device, dtype), "y_vec": vec.to(device, dtype)}


def prepare_inference_condition(z: Tensor, mask_cond: str, ref_list: list | None = None, causal: bool = True):
    """inference.py:283-351: masks [B,1,T,H,W] and masked reference latents [B,C,T,H,W] for the i2v / v2v conditions."""
    B, C, T, H, W = z.shape
    masks = torch.zeros(B, 1, T, H, W)
    masked_z = torch.zeros(B, C, T, H, W)
    if ref_list is None:
        assert mask_cond == "t2v", f"reference is required for {mask_cond}"
    for i in range(B):
        ref = ref_list[i]
        if ref is None and mask_cond != "t2v":
            print("no reference found. will default to cond_type t2v!")
        if ref is not None and T > 1:
            if mask_cond == "i2v_head":
                masks[i, :, 0] = 1
                masked_z[i, :, 0] = ref[0][:, 0]
            elif mask_cond == "i2v_tail":
                masks[i, :, -1] = 1
                masked_z[i, :, -1] = ref[-1][:, -1]
            elif mask_cond in ("v2v_head", "v2v_head_easy"):
                k = (8 if mask_cond == "v2v_head" else 16) + int(causal)
                masks[i, :, :k] = 1
                masked_z[i, :, :k] = ref[0][:, :k]
            elif mask_cond in ("v2v_tail", "v2v_tail_easy"):
                k = (8 if mask_cond == "v2v_tail" else 16) + int(causal)
                masks[i, :, -k:] = 1
                masked_z[i, :, -k:] = ref[0][:, -k:]
            elif mask_cond == "i2v_loop":
                masks[i, :, 0] = 1
                masks[i, :, -1] = 1
                masked_z[i, :, 0] = ref[0][:, 0]
                masked_z[i, :, -1] = ref[-1][:, -1]
            else:
                assert mask_cond == "t2v", f"Unknown mask condition {mask_cond}"
    return masks.to(z.device, z.dtype), masked_z.to(z.device, z.dtype)


def collect_references_batch(reference_paths: list[str], cond_type: str, model_ae, image_size: tuple[int, int],
                             is_causal: bool = False, reader=None) -> list:
    """inference.py:216-280.  `reader(path, image_size, transform_name="resize_crop") -> [C, T, H, W]` is the
    reference's `read_from_path` (media decoding is not part of this library); the frame selection and the VAE
    encodes are the reference's."""
    if reader is None:
        raise ValueError("collect_references_batch needs reader=read_from_path (media decoding is the caller's)")
    p0 = next(model_ae.parameters())
    device, dtype = p0.device, p0.dtype

    def enc(r):
        return model_ae.encode(r.unsqueeze(0).to(device, dtype)).squeeze(0)

    refs_x = []
    for reference_path in reference_paths:
        if reference_path == "":
            refs_x.append(None)
            continue
        ref_path = reference_path.split(";")
        ref = []
        if "v2v" in cond_type:
            r = reader(ref_path[0], image_size, transform_name="resize_crop")
            actual_t = r.size(1)
            target_t = 64 if (actual_t >= 64 and "easy" in cond_type) else 32
            if is_causal:
                target_t += 1
            assert actual_t >= target_t, f"need at least {target_t} reference frames for v2v generation"
            if "head" in cond_type:
                r = r[:, :target_t]
            elif "tail" in cond_type:
                r = r[:, -target_t:]
            else:
                raise NotImplementedError
            ref.append(enc(r))
        elif cond_type == "i2v_head":
            ref.append(enc(reader(ref_path[0], image_size, transform_name="resize_crop")[:, :1]))
        elif cond_type == "i2v_tail":
            ref.append(enc(reader(ref_path[-1], image_size, transform_name="resize_crop")[:, -1:]))
        elif cond_type == "i2v_loop":
            ref.append(enc(reader(ref_path[0], image_size, transform_name="resize_crop")[:, :1]))
            ref.append(enc(reader(ref_path[-1], image_size, transform_name="resize_crop")[:, -1:]))
        else:
            raise NotImplementedError(f"Unknown condition type {cond_type}")
        refs_x.append(ref)
    return refs_x


# ------------------------------------------------------------------------------------------------ denoisers
class DistilledDenoiser:
    """sampling.py:253-288: no CFG; plain Euler steps (the flux image path).  The update runs through the same HIP
    kernel as the CFG sampler with both guidance scales at 1 (v = u2 + (u - u2) + (c - u) = c)."""

    def denoise(self, model, **kwargs) -> Tensor:
        img = kwargs.pop("img")
        timesteps = kwargs.pop("timesteps")
        guidance = kwargs.pop("guidance")
        for k in ("text_osci", "image_osci", "scale_temporal_osci", "flow_shift", "patch_size"):
            kwargs.pop(k, None)
        guidance_vec = torch.full((img.shape[0],), guidance, device=img.device, dtype=img.dtype)
        for t_curr, t_prev in zip(timesteps[:-1], timesteps[1:]):
            t_vec = torch.full((img.shape[0],), t_curr, dtype=img.dtype, device=img.device)
            pred = model(img=img, **kwargs, timesteps=t_vec, guidance=guidance_vec)
            img = img + (t_prev - t_curr) * pred
        return img

    def prepare_guidance(self, text: list[str], optional_models: dict, device, dtype, **kwargs):
        return text, {}


SamplingMethodDict = {SamplingMethod.I2V: I2VDenoiser(), SamplingMethod.DISTILLED: DistilledDenoiser()}


# ------------------------------------------------------------------------------------------------ api
def prepare_api(model, model_ae, model_t5, model_clip, optional_models: dict, reader=None):
    """sampling.py:562-726.  Returns `api_fn(opt, cond_type="t2v", seed=None, sigma_min=1e-5, text=None, neg=None,
    patch_size=2, channel=16, **kwargs) -> video tensor`; `kwargs["ref"]` = list of reference paths (";"-separated)."""

    @torch.inference_mode()
    def api_fn(opt: SamplingOption, cond_type: str = "t2v", seed: int = None, sigma_min: float = 1e-5,
               text: list[str] = None, neg: list[str] = None, patch_size: int = 2, channel: int = 16, **kwargs):
        p0 = next(model.parameters())
        device, dtype = p0.device, p0.dtype
        if seed is None:
            seed = opt.seed if opt.seed is not None else random.randint(0, 2 ** 32 - 1)
        if opt.is_causal_vae:
            num_frames = 1 if opt.num_frames == 1 else (opt.num_frames - 1) // opt.temporal_reduction + 1
        else:
            num_frames = 1 if opt.num_frames == 1 else opt.num_frames // opt.temporal_reduction
        z = get_noise(len(text), opt.height, opt.width, num_frames, device, dtype, seed, patch_size=patch_size,
                      channel=channel // (patch_size ** 2))
        denoiser = SamplingMethodDict[opt.method]
        references = [None] * len(text)
        if cond_type != "t2v" and "ref" in kwargs:
            references = collect_references_batch(kwargs.pop("ref"), cond_type, model_ae, (opt.height, opt.width),
                                                  is_causal=opt.is_causal_vae, reader=reader)
        elif cond_type != "t2v":
            print("your csv file doesn't have a ref column or is not processed properly. will default to cond_type t2v!")
            cond_type = "t2v"
        timesteps = get_schedule(opt.num_steps, (z.shape[-1] * z.shape[-2]) // patch_size ** 2, num_frames,
                                 shift=opt.shift, shift_alpha=opt.flow_shift)
        text, additional_inp = denoiser.prepare_guidance(text=text, optional_models=optional_models, device=device,
                                                         dtype=dtype, neg=neg, guidance_img=opt.guidance_img)
        inp = prepare(model_t5, model_clip, z, prompt=text, patch_size=patch_size)
        inp.update(additional_inp)
        if opt.method in [SamplingMethod.I2V]:
            masks, masked_ref = prepare_inference_condition(z, cond_type, ref_list=references, causal=opt.is_causal_vae)
            inp["masks"] = masks
            inp["masked_ref"] = masked_ref
            inp["sigma_min"] = sigma_min
        x = denoiser.denoise(model, **inp, timesteps=timesteps, guidance=opt.guidance, text_osci=opt.text_osci,
                             image_osci=opt.image_osci,
                             scale_temporal_osci=(opt.scale_temporal_osci and "i2v" in cond_type),
                             flow_shift=opt.flow_shift, patch_size=patch_size)
        x = unpack(x, opt.height, opt.width, num_frames, patch_size=patch_size)
        if cond_type == "i2v_head":
            x[0, :, :1] = references[0][0]
        elif cond_type == "i2v_tail":
            x[0, :, -1:] = references[0][0]
        elif cond_type == "i2v_loop":
            x[0, :, :1] = references[0][0]
            x[0, :, -1:] = references[0][1]
        x = model_ae.decode(x)
        x = x[:, :, : opt.num_frames]
        if not opt.is_causal_vae:   # sampling.py:713-722: `compression` is only read inside the i2v branches
            if cond_type == "i2v_head":
                pad_len = model_ae.compression[0] - 1
                x = x[:, :, pad_len:]
            elif cond_type == "i2v_tail":
                pad_len = model_ae.compression[0] - 1
                x = x[:, :, :-pad_len]
            elif cond_type == "i2v_loop":
                pad_len = model_ae.compression[0] - 1
                x = x[:, :, pad_len:-pad_len]
        return x

    return api_fn
