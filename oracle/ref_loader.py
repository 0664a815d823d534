"""TEST INFRASTRUCTURE ONLY — loader for the reference's own hot-path Python.

This module imports the *unmodified* reference files
    /root/reference/opensora/models/mmdit/{model,layers,math}.py
    /root/reference/opensora/models/hunyuan_vae/{vae,unet_causal_3d_blocks}.py
    /root/reference/opensora/models/vae/utils.py
on CPU, with stubs of our own for the third-party packages that are absent
from this container (flash_attn, liger_kernel, colossalai, diffusers).  The
stubs follow SURVEY.md Appendix D/E; they are restatements of third-party
semantics, not reference code ("reference Python + restated third-party
kernels").

It is used by oracle/make_golden.py (in the build container, where
/root/reference exists) to generate tests/golden/*.npz and by the CPU tests
that pin oracle/*.py against the reference.  It does not exist on the GPU
box (/root/reference is absent there): `available()` returns False and the
callers skip.  Nothing in open_sora_amd/ may import this module.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("OSK_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "opensora/models/mmdit/model.py"))


def _pkg(name: str, path: str | None) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    m.__package__ = name
    sys.modules[name] = m
    return m


def _mod(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_installed = False


def install() -> None:
    """Register synthetic parent packages + third-party stubs (idempotent)."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference tree not found under {REF_ROOT}")
    # the reference mount is writable by root: never leave .pyc files in it
    sys.dont_write_bytecode = True
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")

    import torch
    import torch.nn.functional as F
    from torch import nn

    r = os.path.join(REF_ROOT, "opensora")
    # parent packages resolved by path WITHOUT running the reference __init__.py files
    _pkg("opensora", r)
    _pkg("opensora.models", os.path.join(r, "models"))
    _pkg("opensora.models.mmdit", os.path.join(r, "models/mmdit"))
    _pkg("opensora.models.hunyuan_vae", os.path.join(r, "models/hunyuan_vae"))
    _pkg("opensora.models.vae", os.path.join(r, "models/vae"))
    _pkg("opensora.acceleration", os.path.join(r, "acceleration"))
    _pkg("opensora.utils", None)

    # ---- flash_attn: softmax(q k^T / sqrt(hd)) v on [B, L, H, hd]
    def flash_attn_func(q, k, v, *a, **kw):
        o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
        return o.transpose(1, 2)

    _mod("flash_attn", flash_attn_func=flash_attn_func)

    # ---- liger RMSNorm, "llama" casting, offset 0 (App. E.2)
    class LigerRMSNormFunction:
        @staticmethod
        def apply(x, w, eps, offset, casting_mode, in_place):
            assert casting_mode == "llama"
            xf = x.float()
            y = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype)
            return y * (offset + w)

    # ---- liger RoPE: rotate-half convention, cos/sin [1|B, L, hd] (App. E.3)
    class LigerRopeFunction:
        @staticmethod
        def apply(q, k, cos, sin):
            half = q.shape[-1] // 2
            c = cos[..., :half].unsqueeze(1)
            s = sin[..., :half].unsqueeze(1)

            def rot(x):
                x1, x2 = x[..., :half], x[..., half:]
                return torch.cat((x1 * c - x2 * s, x2 * c + x1 * s), dim=-1).to(x.dtype)

            return rot(q), rot(k)

    _pkg("liger_kernel", None)
    _pkg("liger_kernel.ops", None)
    _mod("liger_kernel.ops.rms_norm", LigerRMSNormFunction=LigerRMSNormFunction)
    _mod("liger_kernel.ops.rope", LigerRopeFunction=LigerRopeFunction)

    _pkg("colossalai", None)
    _mod("colossalai.utils", get_current_device=lambda: torch.device("cpu"))

    class _Registry:
        def register_module(self, *a, **kw):
            return lambda f: f

    _mod("opensora.registry", MODELS=_Registry(), DATASETS=_Registry())
    _mod("opensora.utils.ckpt", load_checkpoint=lambda model, *a, **kw: model)

    # ---- diffusers pieces used by the VAE blocks (App. E.5)
    class Attention(nn.Module):
        """1-head (or C/dim_head heads) attention with GroupNorm, bias, residual."""

        def __init__(self, query_dim, heads=1, dim_head=64, rescale_output_factor=1.0, eps=1e-5,
                     norm_num_groups=32, spatial_norm_dim=None, residual_connection=False, bias=False,
                     upcast_softmax=False, _from_deprecated_attn_block=False, **kw):
            super().__init__()
            inner = heads * dim_head
            self.heads = heads
            self.rescale_output_factor = rescale_output_factor
            self.residual_connection = residual_connection
            self.group_norm = nn.GroupNorm(norm_num_groups, query_dim, eps=eps, affine=True)
            self.to_q = nn.Linear(query_dim, inner, bias=bias)
            self.to_k = nn.Linear(query_dim, inner, bias=bias)
            self.to_v = nn.Linear(query_dim, inner, bias=bias)
            self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])

        def forward(self, hidden_states, attention_mask=None, **kw):
            x = hidden_states
            B, S, C = x.shape
            h = self.group_norm(x.transpose(1, 2)).transpose(1, 2)
            q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
            hd = q.shape[-1] // self.heads

            def split(t):
                return t.view(B, S, self.heads, hd).transpose(1, 2)

            mask = None
            if attention_mask is not None:
                mask = attention_mask.view(B, 1, S, S)
            o = F.scaled_dot_product_attention(split(q), split(k), split(v), attn_mask=mask)
            o = o.transpose(1, 2).reshape(B, S, self.heads * hd)
            o = self.to_out[1](self.to_out[0](o))
            if self.residual_connection:
                o = o + x
            return o / self.rescale_output_factor

    class _Logger:
        def __getattr__(self, n):
            return lambda *a, **k: None

    class BaseOutput(dict):
        pass

    def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
        return torch.randn(shape, generator=generator, device=device, dtype=dtype)

    _pkg("diffusers", None)
    _pkg("diffusers.models", None)
    _mod("diffusers.models.activations", get_activation=lambda name: nn.SiLU())
    _mod("diffusers.models.attention_processor", Attention=Attention)
    lg = _mod("diffusers.utils.logging", get_logger=lambda *a, **k: _Logger())
    du = _pkg("diffusers.utils", None)
    du.logging = lg
    du.BaseOutput = BaseOutput
    _mod("diffusers.utils.torch_utils", randn_tensor=randn_tensor)

    # ---- diffusers scaffolding of AutoencoderKLCausal3D (autoencoder_kl_causal_3d.py:28-49): plain nn.Module,
    # no config registry / hub loading / forward hooks.  Arithmetic-free, so the wrapper's own encode/decode/
    # tiling/blending code runs unmodified.
    class ConfigMixin:
        pass

    class FromOriginalVAEMixin:
        pass

    class ModelMixin(nn.Module):
        pass

    class AttnProcessor:
        pass

    class AttnAddedKVProcessor:
        pass

    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=lambda f: f)
    _mod("diffusers.loaders", FromOriginalVAEMixin=FromOriginalVAEMixin)
    ap = sys.modules["diffusers.models.attention_processor"]
    ap.ADDED_KV_ATTENTION_PROCESSORS = (AttnAddedKVProcessor,)
    ap.CROSS_ATTENTION_PROCESSORS = (AttnProcessor,)
    ap.AttentionProcessor = AttnProcessor
    ap.AttnAddedKVProcessor = AttnAddedKVProcessor
    ap.AttnProcessor = AttnProcessor
    _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    _mod("diffusers.utils.accelerate_utils", apply_forward_hook=lambda f: f)
    _installed = True


def mmdit():
    """-> (model module, layers module, math module) of the reference."""
    install()
    return (
        importlib.import_module("opensora.models.mmdit.model"),
        importlib.import_module("opensora.models.mmdit.layers"),
        importlib.import_module("opensora.models.mmdit.math"),
    )


def hunyuan_ae():
    """-> the reference's autoencoder_kl_causal_3d module (AutoencoderKLCausal3D, AutoEncoder3DConfig)."""
    install()
    return importlib.import_module("opensora.models.hunyuan_vae.autoencoder_kl_causal_3d")


def hunyuan_vae():
    """-> (vae module, unet_causal_3d_blocks module) of the reference."""
    install()
    return (
        importlib.import_module("opensora.models.hunyuan_vae.vae"),
        importlib.import_module("opensora.models.hunyuan_vae.unet_causal_3d_blocks"),
    )


def extract_defs(rel_path: str, names: list[str], namespace: dict) -> dict:
    """TEST INFRASTRUCTURE.  Executes ONLY the named top-level function / class / assignment statements of a reference
    source file inside `namespace` (which supplies their free names: torch, einops, ...).  Used for the reference
    files whose module-level imports (mmengine, peft, colossalai, ...) are absent here: the code that
    runs is the reference's own text, nothing is copied into this repository."""
    import ast

    path = os.path.join(REF_ROOT, rel_path)
    src = open(path).read()
    tree = ast.parse(src)
    want = set(names)
    found = set()
    for node in tree.body:
        tgt = None
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            tgt = node.name
        elif isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            tgt = node.targets[0].id
        if tgt in want:
            code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
            exec(code, namespace)
            found.add(tgt)
    missing = want - found
    if missing:
        raise KeyError(f"{rel_path}: not found: {sorted(missing)}")
    return namespace
