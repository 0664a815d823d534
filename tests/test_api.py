"""open_sora_amd/api.py (the inference API around the hot path) pinned against the reference's
OWN source text: the functions / classes of opensora/utils/sampling.py, opensora/utils/inference.py and
opensora/datasets/aspect.py are executed from /root/reference through oracle.ref_loader.extract_defs (their modules
cannot be imported: mmengine / peft / colossalai are absent).  CPU only; skipped where the reference is not mounted."""
import dataclasses
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


def test_image_sizes_and_sanitize_match_the_reference(ref):
    from open_sora_amd import api

    for res in ["144p", "256px", "360p", "480p", "512px", "720p", "768px", "1024px"]:
        for training in (False, True):
            want = ref["get_aspect_ratios_dict"](ref["get_num_pexels_from_name"](res), training)
            got = api.get_aspect_ratios_dict(api.get_num_pexels_from_name(res), training)
            assert got == want, (res, training)
            for ar in want:
                assert api.get_image_size(res, ar, training) == ref["get_image_size"](res, ar, training)
    assert [f.name for f in dataclasses.fields(api.SamplingOption)] == [f.name for f in dataclasses.fields(ref["SamplingOption"])]
    for f, g in zip(dataclasses.fields(api.SamplingOption), dataclasses.fields(ref["SamplingOption"])):
        if f.name != "method":
            assert f.default == g.default, f.name
    for kw in (dict(resolution="768px", aspect_ratio="16:9", method="i2v"), dict(height=250, width=333, method="distill"),
               dict(resolution="256px", aspect_ratio="9:16", num_frames=129)):
        a = api.sanitize_sampling_option(api.SamplingOption(**kw))
        r = ref["sanitize_sampling_option"](ref["SamplingOption"](**kw))
        assert (a.height, a.width, a.method.value, a.num_frames) == (r.height, r.width, r.method.value, r.num_frames)
    with pytest.raises(AssertionError):
        api.sanitize_sampling_option(api.SamplingOption(resolution="256px"))


@pytest.mark.parametrize("cond", ["t2v", "i2v_head", "i2v_tail", "i2v_loop", "v2v_head", "v2v_tail", "v2v_head_easy", "v2v_tail_easy"])
@pytest.mark.parametrize("causal", [True, False])
def test_prepare_inference_condition_matches_the_reference(ref, cond, causal):
    from open_sora_amd import api

    torch.manual_seed(1)
    z = torch.randn(2, 4, 20, 3, 5).to(BF)
    refs = [[torch.randn(4, 20, 3, 5), torch.randn(4, 20, 3, 5)], None if cond == "t2v" else [torch.randn(4, 20, 3, 5)]]
    if cond == "t2v":
        refs = [None, None]
    m0, r0 = ref["prepare_inference_condition"](z, cond, ref_list=refs, causal=causal)
    m1, r1 = api.prepare_inference_condition(z, cond, ref_list=refs, causal=causal)
    assert torch.equal(m0, m1) and torch.equal(r0, r1) and m1.dtype == z.dtype


class _T5:
    def __call__(self, prompt, added_tokens=0, seq_align=1):
        g = torch.Generator().manual_seed(len(prompt) * 7 + added_tokens % 5)
        return torch.randn(len(prompt), 24, 96, generator=g) * 0.2


class _Clip:
    def __call__(self, prompt):
        g = torch.Generator().manual_seed(len(prompt) + 3)
        return torch.randn(len(prompt), 48, generator=g)


def test_prepare_matches_the_reference(ref):
    from open_sora_amd import api

    z = torch.randn(1, 16, 3, 8, 12).to(BF)
    for prompt in (["a", "b", "c"], "one"):
        a = api.prepare(_T5(), _Clip(), z, prompt=prompt)
        r = ref["prepare"](_T5(), _Clip(), z, prompt=prompt)
        assert set(a) == set(r)
        for k in r:
            assert a[k].dtype == r[k].dtype and torch.equal(a[k], r[k]), k


class _AE(torch.nn.Module):
    """stand-in VAE for the API test (the HIP VAE is tested on the GPU): decode = fixed channel mix, nearest upsample"""
    compression = (4, 8, 8)

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.randn(3, 16) * 0.1)

    def decode(self, z):
        return torch.einsum("oc,bcthw->bothw", self.w.to(z.dtype), z).repeat_interleave(4, 2)

    def encode(self, x):
        return torch.einsum("oc,bcthw->bothw", self.w.t().to(x.dtype), x)[:, :, ::4][:, :, : (x.shape[2] - 1) // 4 + 1]


@pytest.mark.parametrize("cond_type", ["t2v", "i2v_head"])
def test_api_fn_end_to_end_matches_the_reference_api_fn(ref, hip_lib, cond_type):
    """the reference's prepare_api/api_fn (its own text) and ours, driven with the same denoiser module (our MMDiT on
    the CPU emulation of the kernels), the same stand-in text encoders / VAE and the same seed."""
    from open_sora_amd import api, mmdit

    mmdit.set_ops_for_testing(cpu_ops)
    try:
        cfg = dict(configs.GOLDEN["hd64_liger_split"][0], guidance_embed=False)
        model = mmdit.Flux(device_map="cpu", torch_dtype=BF, **cfg)
        model.load_state_dict(torch_params(cfg, dtype=BF), strict=True)
        ae = _AE().to(BF)
        kw = dict(height=64, width=96, num_frames=9, num_steps=3, guidance=7.5, guidance_img=3.0, text_osci=True,
                  image_osci=True, scale_temporal_osci=True, seed=5, is_causal_vae=True, temporal_reduction=4, method="i2v")
        extra = {}
        if cond_type == "i2v_head":
            def reader(path, image_size, transform_name="resize_crop"):
                g = torch.Generator().manual_seed(11)
                return torch.randn(3, 5, image_size[0] // 8, image_size[1] // 8, generator=g)   # "pixels" at the stand-in AE's scale
            extra = dict(ref=["some/path.png"])
            ref["read_from_path"] = reader
        else:
            reader = None
        a_opt = api.sanitize_sampling_option(api.SamplingOption(**kw))
        r_opt = ref["sanitize_sampling_option"](ref["SamplingOption"](**kw))
        ours = api.prepare_api(model, ae, _T5(), _Clip(), {}, reader=reader)(a_opt, cond_type=cond_type, text=["a cat"], channel=64,
                                                                             **dict(extra))
        theirs = ref["prepare_api"](model, ae, _T5(), _Clip(), {})(r_opt, cond_type=cond_type, text=["a cat"], channel=64, **dict(extra))
        assert ours.shape == theirs.shape
        # the reference updates x in bf16 torch ops, ours in one f32 kernel per step: one bf16 rounding apart per step
        scale = max(1.0, theirs.float().abs().max().item())
        assert (ours.float() - theirs.float()).abs().max().item() <= 3e-2 * scale
    finally:
        mmdit.set_ops_for_testing(hip_lib)
