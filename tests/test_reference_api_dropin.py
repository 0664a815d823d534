"""The drop-in at the API level: the REFERENCE'S OWN `prepare_api` / `api_fn`
(opensora/utils/sampling.py:562-726, executed from /root/reference through oracle.ref_loader.extract_defs -- its module
cannot be imported: mmengine / peft / colossalai are absent) drives THIS package's modules unchanged: `MMDiTModel` as
`model`, a VAE with the reference's `encode / decode / compression` interface as `model_ae`.  The package ships no copy of
that host glue (round 2's open_sora_amd/api.py was one and is gone): switching = passing these modules to the reference's
own function.  Checked here on the CPU emulation of the kernels: the reference pipeline runs end to end on our modules, and
its result equals the same pipeline composed from this package's sampler pieces (open_sora_amd.sampling) to one bf16
rounding per Euler step (the reference updates x in bf16 torch ops, ours in one f32 kernel).
CPU only; skipped where the reference is not mounted."""
import math
import os
import random
from dataclasses import dataclass, replace
from enum import Enum

import pytest
import torch

from oracle import configs, ref_loader
from tests import cpu_ops
from tests.util import torch_params

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not mounted")
BF = torch.bfloat16


def _ref_namespace():
    import einops
    from abc import ABC, abstractmethod
    from torch import Tensor, nn

    ns = dict(torch=torch, nn=nn, Tensor=Tensor, math=math, os=os, random=random, dataclass=dataclass, replace=replace,
              Enum=Enum, ABC=ABC, abstractmethod=abstractmethod, rearrange=einops.rearrange, repeat=einops.repeat, HFEmbedder=object, MMDiTModel=object)
    ref_loader.extract_defs("opensora/datasets/aspect.py",
                            ["ASPECT_RATIO_LD_LIST", "get_ratio", "get_aspect_ratios_dict", "get_num_pexels_from_name",
                             "get_image_size"], ns)
    ref_loader.extract_defs("opensora/utils/inference.py", ["SamplingMethod", "prepare_inference_condition",
                                                            "collect_references_batch"], ns)
    ref_loader.extract_defs("opensora/utils/sampling.py",
                            ["SamplingOption", "sanitize_sampling_option", "get_oscillation_gs", "Denoiser", "I2VDenoiser",
                             "DistilledDenoiser", "SamplingMethodDict", "time_shift", "get_res_lin_function", "get_schedule",
                             "get_noise", "pack", "unpack", "prepare", "prepare_api"], ns)
    return ns


@pytest.fixture(scope="module")
def ref():
    return _ref_namespace()


class _T5:
    def __call__(self, prompt, added_tokens=0, seq_align=1):
        g = torch.Generator().manual_seed(len(prompt) * 7 + added_tokens % 5)
        return torch.randn(len(prompt), 24, 96, generator=g) * 0.2


class _Clip:
    def __call__(self, prompt):
        g = torch.Generator().manual_seed(len(prompt) + 3)
        return torch.randn(len(prompt), 48, generator=g)


class _AE(torch.nn.Module):
    """stand-in VAE with the reference interface (the HIP VAE is tested on the GPU): decode = fixed channel mix, nearest upsample"""
    compression = (4, 8, 8)

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.randn(3, 16) * 0.1)

    def decode(self, z):
        return torch.einsum("oc,bcthw->bothw", self.w.to(z.dtype), z).repeat_interleave(4, 2)

    def encode(self, x):
        return torch.einsum("oc,bcthw->bothw", self.w.t().to(x.dtype), x)[:, :, ::4][:, :, : (x.shape[2] - 1) // 4 + 1]


@pytest.mark.parametrize("cond_type", ["t2v", "i2v_head"])
def test_reference_prepare_api_drives_our_modules(ref, hip_lib, cond_type):
    from open_sora_amd import mmdit, sampling

    mmdit.set_ops_for_testing(cpu_ops)
    try:
        cfg = dict(configs.GOLDEN["hd64_liger_split"][0], guidance_embed=False)
        model = mmdit.Flux(device_map="cpu", torch_dtype=BF, **cfg)
        model.load_state_dict(torch_params(cfg, dtype=BF), strict=True)
        ae = _AE().to(BF)
        height, width, frames, steps, seed = 64, 96, 9, 3, 5
        kw = dict(height=height, width=width, num_frames=frames, num_steps=steps, guidance=7.5, guidance_img=3.0, text_osci=True,
                  image_osci=True, scale_temporal_osci=True, seed=seed, is_causal_vae=True, temporal_reduction=4, method="i2v")
        extra, pixels = {}, None
        if cond_type == "i2v_head":
            g = torch.Generator().manual_seed(11)
            pixels = torch.randn(3, 5, height // 8, width // 8, generator=g)   # "pixels" at the stand-in AE's scale
            ref["read_from_path"] = lambda path, image_size, transform_name="resize_crop": pixels.clone()
            extra = dict(ref=["some/path.png"])
        r_opt = ref["sanitize_sampling_option"](ref["SamplingOption"](**kw))
        theirs = ref["prepare_api"](model, ae, _T5(), _Clip(), {})(r_opt, cond_type=cond_type, text=["a cat"], channel=64, **dict(extra))

        # ---- the same pipeline composed from this package's pieces
        T_lat = (frames - 1) // 4 + 1
        with torch.inference_mode():
            z = sampling.get_noise(1, height, width, T_lat, torch.device("cpu"), BF, seed, patch_size=2, channel=16)
            Hl, Wl = z.shape[-2:]
            n_img = T_lat * (Hl // 2) * (Wl // 2)
            text3 = ["a cat", "", ""]
            txt, y_vec = _T5()(text3, added_tokens=n_img).to(BF), _Clip()(text3).to(BF)
            img_ids, txt_ids = sampling.prepare_ids(3, T_lat, Hl, Wl, txt.shape[1], "cpu", BF)
            masks, masked_ref = torch.zeros(1, 1, T_lat, Hl, Wl, dtype=BF), torch.zeros(1, 16, T_lat, Hl, Wl, dtype=BF)
            lat_ref = None
            if cond_type == "i2v_head":
                lat_ref = ae.encode(pixels[None].to(BF))
                masks[:, :, 0] = 1
                masked_ref[:, :, 0] = lat_ref[:, :, 0]
            x = sampling.I2VDenoiser().denoise(
                model, img=sampling.pack(z).repeat(3, 1, 1), timesteps=sampling.get_schedule(steps, (Hl // 2) * (Wl // 2), T_lat),
                guidance=7.5, guidance_img=3.0, masks=masks, masked_ref=masked_ref, text_osci=True, image_osci=True,
                scale_temporal_osci="i2v" in cond_type, img_ids=img_ids, txt=txt, txt_ids=txt_ids, y_vec=y_vec)
            x = sampling.unpack(x, height, width, T_lat)
            if cond_type == "i2v_head":
                x[0, :, :1] = lat_ref[0, :, :1]
            ours = ae.decode(x)[:, :, :frames]
        assert ours.shape == theirs.shape and torch.isfinite(theirs.float()).all()
        scale = max(1.0, theirs.float().abs().max().item())
        assert (ours.float() - theirs.float()).abs().max().item() <= 3e-2 * scale
    finally:
        mmdit.set_ops_for_testing(hip_lib)
