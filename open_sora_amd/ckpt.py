"""Checkpoint loading for the drop-in modules (the step before the hot path: SURVEY.md §8f rank 3).

Mirrors /root/reference/opensora/utils/ckpt.py:84-140 `load_checkpoint` (same arguments; safetensors and .pt/.pth
files; the ColossalAI sharded-directory form is refused: ColossalAI is not part of this library) and adds the
RoPE-convention weight transform the reference ships as a separate checkpoint ("flux1-dev-fused-rope",
docs/train.md:112): `use_liger_rope=True` rotates pairs (j, j + hd/2), the eager path pairs (2j, 2j+1); the two
models are the same function iff the q / k projection output features (and biases, and the QK-norm scales) of every
head are permuted by `rearrange_tensor`'s index map (mmdit/math.py:68-91: new[d] = old[2d], new[hd/2 + d] = old[2d+1]).
State-dict keys are the reference's, so `Open_Sora_v2.safetensors` / `hunyuan_vae.safetensors` load unchanged.
"""
from __future__ import annotations

import os
import re

import torch
from torch import Tensor, nn


def print_load_warning(missing: list[str], unexpected: list[str]) -> None:
    """ckpt.py:65-81 (log to stdout instead of the reference's logger)."""
    if missing:
        print(f"Got {len(missing)} missing keys:\n\t" + "\n\t".join(missing))
    if unexpected:
        print(f"Got {len(unexpected)} unexpected keys:\n\t" + "\n\t".join(unexpected))
    if not missing and not unexpected:
        print("Model loaded successfully")


def load_checkpoint(model: nn.Module, path: str, cache_dir: str = None, device_map: torch.device | str = "cpu",
                    cai_model_name: str = "model", strict: bool = False, rename_keys: dict = None) -> nn.Module:
    """ckpt.py:84-140.  No network here: a path that does not exist is an error (the reference would try the
    Hugging Face hub)."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"Could not find checkpoint at {path} (no hub download in this library)")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file

        ckpt = load_file(path, device="cpu")
        if rename_keys is not None:  # ckpt.py:117-128: first matching old prefix wins
            renamed = {}
            for old_key, v in ckpt.items():
                new_key = old_key
                for old_prefix, new_prefix in rename_keys.items():
                    if old_prefix in old_key:
                        new_key = old_key.replace(old_prefix, new_prefix)
                        break
                renamed[new_key] = v
            ckpt = renamed
    elif path.endswith(".pt") or path.endswith(".pth"):
        ckpt = torch.load(path, map_location=device_map)
    else:
        raise ValueError(f"Invalid checkpoint path: {path} (ColossalAI sharded directories are not supported; "
                         f"cai_model_name={cai_model_name!r})")
    missing, unexpected = model.load_state_dict(ckpt, strict=strict)
    print_load_warning(list(missing), list(unexpected))
    return model


# ------------------------------------------------------------------------------------------------ RoPE conventions
def rope_feature_index(head_dim: int, to: str = "half") -> Tensor:
    """index map over one head's features.  to="half": new[d] = old[2d], new[hd/2+d] = old[2d+1]
    (rearrange_tensor, math.py:68-91); to="interleaved": its inverse (reverse_rearrange_tensor, :94-117)."""
    if head_dim % 2:
        raise ValueError("The last dimension D must be even.")
    half = head_dim // 2
    idx = torch.empty(head_dim, dtype=torch.long)
    if to == "half":
        idx[:half] = torch.arange(0, head_dim, 2)
        idx[half:] = torch.arange(1, head_dim, 2)
    elif to == "interleaved":
        idx[::2] = torch.arange(half)
        idx[1::2] = torch.arange(half, head_dim)
    else:
        raise ValueError(to)
    return idx


_QK_SPLIT = re.compile(r"(^|\.)(q_proj|k_proj)\.(weight|bias)$")
_QKV_FUSED = re.compile(r"(^|\.)(img_attn|txt_attn)\.qkv\.(weight|bias)$")
_LINEAR1 = re.compile(r"(^|\.)single_blocks\.\d+\.linear1\.(weight|bias)$")
_QK_SCALE = re.compile(r"\.norm\.(query_norm|key_norm)\.scale$")


def convert_rope_convention(state_dict: dict, hidden_size: int, num_heads: int, to: str = "half") -> dict:
    """Returns a new state dict whose q / k projections (fused `qkv` / `linear1` rows [0, 2D) or split
    `q_proj` / `k_proj`), their biases and the QK-norm scales are permuted per head so that the model evaluated with
    the other RoPE convention (`use_liger_rope` flipped) computes the same function.  v, mlp and every other tensor
    are untouched (attention scores are invariant under a consistent permutation of q's and k's head features)."""
    D, H = hidden_size, num_heads
    hd = D // H
    per_head = rope_feature_index(hd, to)
    rows = (torch.arange(H)[:, None] * hd + per_head[None, :]).reshape(-1)  # permutation of one D-row block
    out = {}
    for key, t in state_dict.items():
        if _QK_SPLIT.search(key):
            out[key] = t.index_select(0, rows.to(t.device))
        elif _QKV_FUSED.search(key) or _LINEAR1.search(key):
            parts = [t[:D].index_select(0, rows.to(t.device)), t[D: 2 * D].index_select(0, rows.to(t.device)), t[2 * D:]]
            out[key] = torch.cat(parts, 0)
        elif _QK_SCALE.search(key):
            out[key] = t.index_select(0, per_head.to(t.device))
        else:
            out[key] = t
    return out
