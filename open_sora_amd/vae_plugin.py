"""Plug-in of the HIP causal-VAE kernels on the REFERENCE's own modules ("VAE plugin point").

The reference swaps parts of its VAE through ColossalAI ShardFormer policies: a `SubModuleReplacementDescription(suffix, target_module)`
names a sub-module and a class whose `from_native_module(module, ...)` builds the replacement from the native one
(/root/reference/opensora/models/hunyuan_vae/policy.py:13-48 lists `EncoderCausal3D` / `DecoderCausal3D` with per-layer targets;
the layers themselves: hunyuan_vae/unet_causal_3d_blocks.py:82-96 CausalConv3d, :184-259 ResnetBlockCausal3D, :262-351 mid block).
This module supplies targets of that shape that run `osk_causal_conv3d_* / osk_groupnorm_* / osk_attention_hd512_*` instead of
torch:

  HipEncoderCausal3D / HipDecoderCausal3D   whole encoder / decoder: NCTHW in, NCTHW out, ONE layout conversion at each end, the
                                            NDHWC engine of open_sora_amd/hunyuan_vae.py in between (GroupNorm statistics in the conv
                                            epilogues, upsample folded into the conv, hd-512 flash attention in the mid block);
  HipCausalConv3d                           one CausalConv3d layer (policy.py's per-layer granularity): converts layout per call -- the
                                            drop-in for a maintainer who swaps single layers; use the whole-module targets for speed.

A replacement ADOPTS the native module's sub-modules, parameters and buffers (same objects): the state dict, `.to()`, checkpoint
loading and the reference's own `AutoencoderKLCausal3D.encode / decode / tiled_* / blend_*` Python keep working unchanged around it
(the counterpart of HipDoubleStreamBlockProcessor on the denoiser side).  `install(ae)` does the two replacements on a reference
`AutoencoderKLCausal3D`; `uninstall(ae)` puts the native modules back.  Inference only (no autograd through the kernels).
"""
from __future__ import annotations

import torch
from torch import Tensor, nn

from . import hunyuan_vae as _hv


class _Adopted(nn.Module):
    """replacement that shares the native module's children / parameters / buffers and plain attributes"""

    def __init__(self, native: nn.Module):
        super().__init__()
        self.__dict__["_osk_native"] = native
        for k, v in native.__dict__.items():
            if k in ("_modules", "_parameters", "_buffers"):
                self.__dict__[k] = v                       # the SAME dicts: a later load_state_dict / .to() reaches both
            elif not k.startswith("_") and k != "training":
                self.__dict__.setdefault(k, v)             # layers_per_block, add_attention, upsample_factor, ...
        self.training = native.training

    @classmethod
    def from_native_module(cls, module: nn.Module, *args, **kwargs):
        """ShardFormer's constructor convention (policy.py: SubModuleReplacementDescription.target_module)"""
        _assert_supported(module)
        return cls(module)

    def native_module(self) -> nn.Module:
        return self.__dict__["_osk_native"]


def _assert_supported(native: nn.Module) -> None:
    """The NDHWC engine hard-codes what the shipped Hunyuan VAE uses: swish, replicate causal padding, dilation 1, stride in {1, 2},
    output_scale_factor 1, no dropout, GroupNorm(32, eps 1e-6), upsample factors in {1, 2}.  The reference's classes are
    configurable beyond that (unet_causal_3d_blocks.py:63-96 `pad_mode`, :184-259 `output_scale_factor`, `dropout`, `non_linearity`):
    adopting a module configured otherwise would change its numerics silently, so it is refused here."""
    def bad(what, m):
        raise ValueError(f"open_sora_amd VAE plug-in: unsupported configuration ({what}) on {type(m).__name__}; "
                         "the HIP engine implements the shipped Hunyuan VAE settings only")

    for m in native.modules():
        if hasattr(m, "pad_mode") and getattr(m, "pad_mode") != "replicate":
            bad(f"pad_mode={m.pad_mode!r}", m)
        if isinstance(m, nn.Conv3d):
            if tuple(m.dilation) != (1, 1, 1) or m.groups != 1:
                bad(f"dilation={tuple(m.dilation)}, groups={m.groups}", m)
            if any(s_ not in (1, 2) for s_ in m.stride) or any(k_ not in (1, 3) for k_ in m.kernel_size):
                bad(f"stride={tuple(m.stride)}, kernel={tuple(m.kernel_size)}", m)
        if hasattr(m, "output_scale_factor") and float(getattr(m, "output_scale_factor")) != 1.0:
            bad(f"output_scale_factor={m.output_scale_factor}", m)
        if isinstance(m, nn.Dropout) and m.p != 0.0 and native.training:
            bad(f"dropout p={m.p} in training mode", m)
        nl = getattr(m, "nonlinearity", None)
        if isinstance(nl, nn.Module) and not isinstance(nl, nn.SiLU):
            bad(f"nonlinearity={type(nl).__name__}", m)
        if isinstance(m, nn.GroupNorm) and not m.affine:
            bad("GroupNorm without affine parameters", m)
        uf = getattr(m, "upsample_factor", None)
        if uf is not None and any(int(f) not in (1, 2) for f in (uf if isinstance(uf, (tuple, list)) else (uf,))):
            bad(f"upsample_factor={uf}", m)


def _io_dtype(x: Tensor):
    if torch.is_grad_enabled() and x.requires_grad:
        raise RuntimeError("open_sora_amd VAE plug-in: inference only (the HIP kernels carry no autograd)")
    return x.dtype


class HipEncoderCausal3D(_Adopted):
    """EncoderCausal3D.forward (hunyuan_vae/vae.py:128-150): [B, C, T, H, W] -> [B, 2 z, T', H', W'] on the HIP kernels."""

    def forward(self, sample: Tensor) -> Tensor:
        assert sample.ndim == 5, "The input tensor should have 5 dimensions"
        dt = _io_dtype(sample)
        h = _hv.run_encoder(self, _hv._to_ndhwc(sample, _hv._pad8(sample.shape[1])))
        return _hv._to_ncthw(h, dt)


class HipDecoderCausal3D(_Adopted):
    """DecoderCausal3D.forward (hunyuan_vae/vae.py:245-277): [B, z, T', H', W'] -> [B, C, T, H, W] on the HIP kernels."""

    def forward(self, sample: Tensor) -> Tensor:
        assert sample.ndim == 5, "The input tensor should have 5 dimensions."
        dt = _io_dtype(sample)
        h = _hv.run_decoder(self, _hv._to_ndhwc(sample, _hv._pad8(sample.shape[1])))
        return _hv._to_ncthw(h, dt)


class HipCausalConv3d(_Adopted):
    """CausalConv3d.forward (unet_causal_3d_blocks.py:92-96: replicate / causal padding + Conv3d) as one osk_causal_conv3d launch
    between two layout conversions."""

    def forward(self, x: Tensor) -> Tensor:
        dt = _io_dtype(x)
        return _hv._to_ncthw(_hv._conv(self, _hv._to_ndhwc(x, _hv._pad8(x.shape[1]))), dt)


def install(ae: nn.Module) -> nn.Module:
    """Replace `ae.encoder` / `ae.decoder` of a reference AutoencoderKLCausal3D (autoencoder_kl_causal_3d.py:84) by the HIP targets.
    Returns `ae`.  quant_conv / post_quant_conv (1 x 1 x 1 over 16 / 32 channels), the posterior, tiling and blending stay the
    reference's own Python."""
    if not isinstance(ae.encoder, HipEncoderCausal3D):
        ae.encoder = HipEncoderCausal3D.from_native_module(ae.encoder)
    if not isinstance(ae.decoder, HipDecoderCausal3D):
        ae.decoder = HipDecoderCausal3D.from_native_module(ae.decoder)
    return ae


def uninstall(ae: nn.Module) -> nn.Module:
    for name in ("encoder", "decoder"):
        m = getattr(ae, name)
        if isinstance(m, _Adopted):
            setattr(ae, name, m.native_module())
    return ae
