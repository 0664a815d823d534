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


def make_params(shapes: dict, seed: int = 0, round_bf16: bool = True, workers: int = 1) -> dict:
    """name -> fp32 numpy array.  Weights ~ N(0, 1/fan_in), biases ~ N(0, 0.02^2),
    RMSNorm scales ~ 1 + 0.1 N(0,1), GroupNorm weight ~ 1 + 0.1 N, all bf16-representable.
    Every tensor is a pure function of (name, seed, shape), so `workers` > 1 only spreads the tensors over threads
    (numpy releases the GIL inside its ufuncs): same bits, the 0.8 G parameters of the XL denoiser in seconds on a many-core host."""

    def one(item):
        name, shape = item
        if name.endswith(".scale") or (name.endswith(".weight") and len(shape) == 1):
            a = normal(name, seed, shape, std=0.1, mean=1.0)
        elif name.endswith(".bias"):
            a = normal(name, seed, shape, std=0.02)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = normal(name, seed, shape, std=fan_in ** -0.5)
        return name, (bf16_round(a) if round_bf16 else a)

    if workers and workers > 1:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=int(workers)) as pool:
            return dict(pool.map(one, shapes.items()))
    return dict(map(one, shapes.items()))


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


# --------------------------------------------------------------------------- Hunyuan causal 3-D VAE weights
def vae_param_shapes(cfg: dict) -> dict:
    """Parameter name -> shape of the reference AutoencoderKLCausal3D state dict
    (opensora/models/hunyuan_vae/autoencoder_kl_causal_3d.py:99-133; vae.py:58-119,158-231;
    unet_causal_3d_blocks.py:92,216-245,312-341).  cfg keys: in_channels, out_channels, latent_channels,
    block_out_channels, layers_per_block (AutoEncoder3DConfig field names)."""
    ch = list(cfg["block_out_channels"])
    lpb = cfg.get("layers_per_block", 2)
    zc = cfg.get("latent_channels", 16)
    cin, cout = cfg.get("in_channels", 3), cfg.get("out_channels", 3)
    s: dict = {}

    def conv(name, co, ci, k):
        s[name + ".weight"] = (co, ci, k, k, k)
        s[name + ".bias"] = (co,)

    def norm(name, c):
        s[name + ".weight"] = (c,)
        s[name + ".bias"] = (c,)

    def resnet(name, ci, co):
        norm(name + ".norm1", ci)
        conv(name + ".conv1.conv", co, ci, 3)
        norm(name + ".norm2", co)
        conv(name + ".conv2.conv", co, co, 3)
        if ci != co:
            conv(name + ".conv_shortcut.conv", co, ci, 1)

    def mid(name, c):
        a = name + ".attentions.0"
        norm(a + ".group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            s[f"{a}.{n}.weight"] = (c, c)
            s[f"{a}.{n}.bias"] = (c,)
        resnet(name + ".resnets.0", c, c)
        resnet(name + ".resnets.1", c, c)

    # encoder (vae.py:58-119)
    conv("encoder.conv_in.conv", ch[0], cin, 3)
    prev = ch[0]
    for i, c in enumerate(ch):
        for j in range(lpb):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
        if i < len(ch) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv.conv", c, c, 3)
        prev = c
    mid("encoder.mid_block", ch[-1])
    norm("encoder.conv_norm_out", ch[-1])
    conv("encoder.conv_out.conv", 2 * zc, ch[-1], 3)
    # decoder (vae.py:158-231)
    conv("decoder.conv_in.conv", ch[-1], zc, 3)
    rev = ch[::-1]
    prev = rev[0]
    for i, c in enumerate(rev):
        for j in range(lpb + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
        if i < len(ch) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv.conv", c, c, 3)
        prev = c
    mid("decoder.mid_block", ch[-1])
    norm("decoder.conv_norm_out", ch[0])
    conv("decoder.conv_out.conv", cout, ch[0], 3)
    s["quant_conv.weight"] = (2 * zc, 2 * zc, 1, 1, 1)
    s["quant_conv.bias"] = (2 * zc,)
    s["post_quant_conv.weight"] = (zc, zc, 1, 1, 1)
    s["post_quant_conv.bias"] = (zc,)
    return s


def vae_video(B: int, T: int, H: int, W: int, seed: int = 7, C: int = 3) -> np.ndarray:
    """Synthetic pixel video in [-1, 1]-ish range, NCTHW, bf16-representable."""
    return bf16_round(np.clip(normal("in.video", seed, (B, C, T, H, W), std=0.5), -1.0, 1.0))


def vae_latent(B: int, T: int, h: int, w: int, seed: int = 8, C: int = 16) -> np.ndarray:
    """Synthetic (already scaled) latent, NCTHW, bf16-representable."""
    return bf16_round(normal("in.latent", seed, (B, C, T, h, w), std=0.5))
