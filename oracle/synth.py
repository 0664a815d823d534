"""TEST INFRASTRUCTURE ONLY — deterministic synthetic tensors.

A counter-based generator (splitmix64 -> Box-Muller) in plain numpy, so the
same (name, seed, shape) gives bit-identical fp32 values in the build
container (where goldens are made from the reference) and on the GPU box
(where the HIP path is checked against oracle + goldens) regardless of the
numpy/torch version.  There are no real checkpoints offline,
so every parity statement in this repo is on these synthetic weights.
"""
from __future__ import annotations

import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(name: str, seed: int, n: int) -> np.ndarray:
    """n float64 values in (0,1), a pure function of (name, seed, index)."""
    key = np.uint64((zlib.crc32(name.encode("utf-8")) * 0x100000001B3 + seed) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + key) & _M64
    bits = _splitmix64(_splitmix64(ctr))
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(name: str, seed: int, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    m = (n + 1) // 2
    u = uniform01(name, seed, 2 * m)
    r = np.sqrt(-2.0 * np.log(u[:m]))
    th = 2.0 * np.pi * u[m:]
    z = np.concatenate([r * np.cos(th), r * np.sin(th)])[:n]
    return (mean + std * z).astype(np.float32).reshape(shape)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round fp32 -> nearest-even bf16, returned as fp32 (so weights are exactly bf16-representable)."""
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    lsb = (b >> np.uint32(16)) & np.uint32(1)
    r = (b + np.uint32(0x7FFF) + lsb) & np.uint32(0xFFFF0000)
    return r.view(np.float32).reshape(x.shape)


# --------------------------------------------------------------------------- MMDiT weights
def mmdit_param_shapes(cfg: dict) -> dict:
    """Parameter name -> shape, following the reference state-dict
    (opensora/models/mmdit/model.py:98-137, layers.py:138-152,179-184,256-293,337-376,391-396)."""
    D = cfg["hidden_size"]
    H = cfg["num_heads"]
    hd = D // H
    R = int(D * cfg["mlp_ratio"])
    C = cfg["in_channels"]
    p2 = cfg.get("patch_size", 2) ** 2
    s: dict = {}

    def lin(name, n_out, n_in, bias=True):
        s[name + ".weight"] = (n_out, n_in)
        if bias:
            s[name + ".bias"] = (n_out,)

    lin("img_in", D, C)
    for e, n_in in (("time_in", 256), ("vector_in", cfg["vec_in_dim"])):
        lin(e + ".in_layer", D, n_in)
        lin(e + ".out_layer", D, D)
    if cfg.get("guidance_embed", False):
        lin("guidance_in.in_layer", D, 256)
        lin("guidance_in.out_layer", D, D)
    if cfg.get("cond_embed", False):
        lin("cond_in", D, C + p2)
    lin("txt_in", D, cfg["context_in_dim"])
    fused = cfg.get("fused_qkv", True)
    qb = cfg.get("qkv_bias", False)
    for i in range(cfg["depth"]):
        for st in ("img", "txt"):
            b = f"double_blocks.{i}.{st}"
            lin(b + "_mod.lin", 6 * D, D)
            if fused:
                lin(b + "_attn.qkv", 3 * D, D, qb)
            else:
                for n in ("q_proj", "k_proj", "v_proj"):
                    lin(f"{b}_attn.{n}", D, D, qb)
            s[b + "_attn.norm.query_norm.scale"] = (hd,)
            s[b + "_attn.norm.key_norm.scale"] = (hd,)
            lin(b + "_attn.proj", D, D)
            lin(b + "_mlp.0", R, D)
            lin(b + "_mlp.2", D, R)
    for i in range(cfg["depth_single_blocks"]):
        b = f"single_blocks.{i}"
        if fused:
            lin(b + ".linear1", 3 * D + R, D)
        else:
            lin(b + ".q_proj", D, D)
            lin(b + ".k_proj", D, D)
            lin(b + ".v_mlp", D + R, D)
        lin(b + ".linear2", D, D + R)
        s[b + ".norm.query_norm.scale"] = (hd,)
        s[b + ".norm.key_norm.scale"] = (hd,)
        lin(b + ".modulation.lin", 3 * D, D)
    lin("final_layer.linear", C, D)  # LastLayer(hidden, patch_size=1, out_channels=in_channels), model.py:137
    lin("final_layer.adaLN_modulation.1", 2 * D, D)
    return s


def make_params(shapes: dict, seed: int = 0, round_bf16: bool = True) -> dict:
    """name -> fp32 numpy array.  Weights ~ N(0, 1/fan_in), biases ~ N(0, 0.02^2),
    RMSNorm scales ~ 1 + 0.1 N(0,1), GroupNorm weight ~ 1 + 0.1 N, all bf16-representable."""
    out = {}
    for name, shape in shapes.items():
        if name.endswith(".scale") or (name.endswith(".weight") and len(shape) == 1):
            a = normal(name, seed, shape, std=0.1, mean=1.0)
        elif name.endswith(".bias"):
            a = normal(name, seed, shape, std=0.02)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = normal(name, seed, shape, std=fan_in ** -0.5)
        out[name] = bf16_round(a) if round_bf16 else a
    return out


def mmdit_inputs(cfg: dict, B: int, T: int, h: int, w: int, L_txt: int, seed: int = 42, t: float = 0.7) -> dict:
    """Synthetic forward() inputs shaped like opensora/utils/sampling.py:401-459 builds them
    (img packed [B, T*h*w, 64]; ids = (t,h,w) grid; txt_ids = 0)."""
    C = cfg["in_channels"]
    p2 = cfg.get("patch_size", 2) ** 2
    L = T * h * w
    ids = np.zeros((T, h, w, 3), np.float32)
    ids[..., 0] = np.arange(T)[:, None, None]
    ids[..., 1] = np.arange(h)[None, :, None]
    ids[..., 2] = np.arange(w)[None, None, :]
    d = {
        "img": bf16_round(normal("in.img", seed, (B, L, C))),
        "img_ids": np.broadcast_to(ids.reshape(1, L, 3), (B, L, 3)).copy(),
        "txt": bf16_round(normal("in.txt", seed + 1, (B, L_txt, cfg["context_in_dim"]), std=0.2)),
        "txt_ids": np.zeros((B, L_txt, 3), np.float32),
        "timesteps": np.full((B,), t, np.float32),
        "y_vec": bf16_round(normal("in.y", seed + 1, (B, cfg["vec_in_dim"]))),
    }
    if cfg.get("cond_embed", False):
        d["cond"] = bf16_round(normal("in.cond", seed + 2, (B, L, C + p2)))
    if cfg.get("guidance_embed", False):
        d["guidance"] = np.full((B,), 4.0, np.float32)
    return d
