"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference MMDiT denoiser.

A functional (state-dict driven) restatement, in plain torch CPU ops, of
    /root/reference/opensora/models/mmdit/model.py   (MMDiTModel.prepare_block_inputs / forward_ckpt)
    /root/reference/opensora/models/mmdit/layers.py  (processors, Modulation, QKNorm, LastLayer, embedders)
    /root/reference/opensora/models/mmdit/math.py    (attention, rope, liger_rope, apply_rope)
plus the third-party pieces the reference only imports (flash-attn, Liger
RMSNorm / RoPE).  Each function cites the lines it follows.

Pinned: tests/test_oracle_vs_reference.py runs this file against the real
reference Python (oracle/ref_loader.py) when /root/reference is mounted, and
tests/test_oracle_golden.py against the committed tests/golden/mmdit_*.npz made
from that reference by oracle/make_golden.py.

Run in fp32 it is the truth for parity tests; run with bf16 tensors it keeps
the reference's rounding points and serves as the "reference eager bf16"
comparator of SURVEY.md §8(d).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product path
(open_sora_amd/) never does.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _lin(sd: dict, name: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


# ----------------------------------------------------------------------------- embedders
def timestep_embedding(t: Tensor, dim: int = 256, max_period: int = 10000, time_factor: float = 1000.0) -> Tensor:
    """layers.py:68-88 — [cos | sin] of 1000*t*exp(-ln(1e4) i/half), fp32, cast to t.dtype."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = (time_factor * t)[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    return emb.to(t.dtype) if torch.is_floating_point(t) else emb


def mlp_embedder(sd: dict, name: str, x: Tensor) -> Tensor:
    """layers.py:91-99 — Linear -> SiLU -> Linear."""
    return _lin(sd, name + ".out_layer", F.silu(_lin(sd, name + ".in_layer", x)))


def rope_angles(ids: Tensor, axes_dim, theta: float) -> Tensor:
    """Per-axis angles pos * theta^(-2i/d_axis), axes concatenated along the pair index
    (math.py:39-47,50-57; layers.py:38-44,55-63).  ids [B, L, n_axes] -> [B, L, hd/2] fp64."""
    outs = []
    for a, d in enumerate(axes_dim):
        scale = torch.arange(0, d, 2, dtype=torch.float64) / d
        omega = 1.0 / (theta ** scale)
        outs.append(ids[..., a].double()[..., None] * omega)
    return torch.cat(outs, dim=-1)


def rms_norm(x: Tensor, scale: Tensor, eps: float = 1e-6) -> Tensor:
    """layers.py:107-111 (== Liger "llama" mode): fp32 normalise, cast back, * scale."""
    xf = x.float()
    rrms = torch.rsqrt(torch.mean(xf * xf, dim=-1, keepdim=True) + eps)
    return (xf * rrms).to(x.dtype) * scale


def apply_rope_interleaved(x: Tensor, ang: Tensor) -> Tensor:
    """math.py:50-65 — pairs (2j, 2j+1), fp32 rotation, cast back.  x [B,H,L,hd], ang [B,L,hd/2]."""
    xf = x.float().reshape(*x.shape[:-1], -1, 2)
    c = torch.cos(ang).float()[:, None]
    s = torch.sin(ang).float()[:, None]
    o0 = c * xf[..., 0] - s * xf[..., 1]
    o1 = s * xf[..., 0] + c * xf[..., 1]
    return torch.stack([o0, o1], dim=-1).reshape(x.shape).to(x.dtype)


def apply_rope_half(x: Tensor, ang: Tensor) -> Tensor:
    """Liger rotate-half convention: pairs (j, j+hd/2); cos/sin fp32 of fp32 angles
    (math.py:39-47 computes omega, angles, cos and sin all in fp32)."""
    half = x.shape[-1] // 2
    a32 = ang.float()
    c = torch.cos(a32)[:, None]
    s = torch.sin(a32)[:, None]
    x1, x2 = x[..., :half], x[..., half:]
    return torch.cat((x1 * c - x2 * s, x2 * c + x1 * s), dim=-1).to(x.dtype)


def rope_angles_liger(ids: Tensor, axes_dim, theta: float) -> Tensor:
    """math.py:39-47 — fp32 omega and fp32 einsum (liger path)."""
    outs = []
    for a, d in enumerate(axes_dim):
        scale = torch.arange(0, d, 2, dtype=torch.float32) / d
        omega = 1.0 / (theta ** scale)
        outs.append(ids[..., a].float()[..., None] * omega)
    return torch.cat(outs, dim=-1)


def sdpa(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """flash_attn_func semantics: softmax(q k^T hd^-1/2) v, non-causal.
    q,k,v [B,H,L,hd] -> [B,L,H*hd].  Softmax in fp32 regardless of the I/O dtype."""
    B, H, L, hd = q.shape
    if B * H * L * L > (1 << 28):  # bound the score matrix to one (b, h) at a time at large L
        o = torch.empty(B, H, L, hd, dtype=q.dtype)
        for b in range(B):
            for h in range(H):
                s = torch.matmul(q[b, h].float(), k[b, h].float().t()) * (hd ** -0.5)
                o[b, h] = torch.matmul(torch.softmax(s, dim=-1), v[b, h].float()).to(q.dtype)
    else:
        s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * (hd ** -0.5)
        p = torch.softmax(s, dim=-1)
        o = torch.matmul(p, v.float()).to(q.dtype)
    return o.permute(0, 2, 1, 3).reshape(B, L, H * hd)


def attention(q: Tensor, k: Tensor, v: Tensor, ang: Tensor, rope_mode: str) -> Tensor:
    """math.py:22-36."""
    rope = apply_rope_interleaved if rope_mode == "interleaved" else apply_rope_half
    return sdpa(rope(q, ang), rope(k, ang), v)


# ----------------------------------------------------------------------------- blocks
def _layer_norm(x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), eps=1e-6)


def _modulation(sd: dict, name: str, vec: Tensor, n: int):
    """layers.py:179-192 — Linear(SiLU(vec))[:, None, :].chunk(n)."""
    return _lin(sd, name + ".lin", F.silu(vec))[:, None, :].chunk(n, dim=-1)


def _heads(x: Tensor, H: int) -> Tensor:
    B, L, D = x.shape
    return x.view(B, L, H, D // H).permute(0, 2, 1, 3)


def _qkv(sd: dict, prefix: str, x: Tensor, H: int, fused: bool):
    """layers.py:208-220 / 225-236: q,k,v [B,H,L,hd] with QK-RMSNorm (q,k cast to v.dtype)."""
    if fused:
        q, k, v = _lin(sd, prefix + ".qkv", x).chunk(3, dim=-1)
    else:
        q, k, v = (_lin(sd, f"{prefix}.{n}", x) for n in ("q_proj", "k_proj", "v_proj"))
    q, k, v = _heads(q, H), _heads(k, H), _heads(v, H)
    q = rms_norm(q, sd[prefix + ".norm.query_norm.scale"]).to(v.dtype)
    k = rms_norm(k, sd[prefix + ".norm.key_norm.scale"]).to(v.dtype)
    return q, k, v


def double_block(sd: dict, cfg: dict, i: int, img: Tensor, txt: Tensor, vec: Tensor, ang: Tensor, rope_mode: str):
    """DoubleStreamBlockProcessor.__call__, layers.py:195-253."""
    H = cfg["num_heads"]
    fused = cfg.get("fused_qkv", True)
    b = f"double_blocks.{i}"
    i_sh1, i_sc1, i_g1, i_sh2, i_sc2, i_g2 = _modulation(sd, b + ".img_mod", vec, 6)
    t_sh1, t_sc1, t_g1, t_sh2, t_sc2, t_g2 = _modulation(sd, b + ".txt_mod", vec, 6)

    iq, ik, iv = _qkv(sd, b + ".img_attn", (1 + i_sc1) * _layer_norm(img) + i_sh1, H, fused)
    tq, tk, tv = _qkv(sd, b + ".txt_attn", (1 + t_sc1) * _layer_norm(txt) + t_sh1, H, fused)
    a = attention(torch.cat((tq, iq), 2), torch.cat((tk, ik), 2), torch.cat((tv, iv), 2), ang, rope_mode)
    Lt = txt.shape[1]
    t_a, i_a = a[:, :Lt], a[:, Lt:]

    def mlp(st, x):
        return _lin(sd, f"{b}.{st}_mlp.2", F.gelu(_lin(sd, f"{b}.{st}_mlp.0", x), approximate="tanh"))

    img = img + i_g1 * _lin(sd, b + ".img_attn.proj", i_a)
    img = img + i_g2 * mlp("img", (1 + i_sc2) * _layer_norm(img) + i_sh2)
    txt = txt + t_g1 * _lin(sd, b + ".txt_attn.proj", t_a)
    txt = txt + t_g2 * mlp("txt", (1 + t_sc2) * _layer_norm(txt) + t_sh2)
    return img, txt


def single_block(sd: dict, cfg: dict, i: int, x: Tensor, vec: Tensor, ang: Tensor, rope_mode: str) -> Tensor:
    """SingleStreamBlockProcessor.__call__, layers.py:309-334."""
    H = cfg["num_heads"]
    D = cfg["hidden_size"]
    b = f"single_blocks.{i}"
    shift, scale, gate = _modulation(sd, b + ".modulation", vec, 3)
    xm = (1 + scale) * _layer_norm(x) + shift
    if cfg.get("fused_qkv", True):
        y = _lin(sd, b + ".linear1", xm)
        q, k, v, mlp = y[..., :D], y[..., D : 2 * D], y[..., 2 * D : 3 * D], y[..., 3 * D :]
    else:
        q = _lin(sd, b + ".q_proj", xm)
        k = _lin(sd, b + ".k_proj", xm)
        vm = _lin(sd, b + ".v_mlp", xm)
        v, mlp = vm[..., :D], vm[..., D:]
    q, k, v = _heads(q, H), _heads(k, H), _heads(v, H)
    q = rms_norm(q, sd[b + ".norm.query_norm.scale"]).to(v.dtype)
    k = rms_norm(k, sd[b + ".norm.key_norm.scale"]).to(v.dtype)
    a = attention(q, k, v, ang, rope_mode)
    out = _lin(sd, b + ".linear2", torch.cat((a, F.gelu(mlp, approximate="tanh")), 2))
    return x + gate * out


def last_layer(sd: dict, x: Tensor, vec: Tensor) -> Tensor:
    """LastLayer.forward, layers.py:398-402 — note (shift, scale) order."""
    shift, scale = _lin(sd, "final_layer.adaLN_modulation.1", F.silu(vec)).chunk(2, dim=1)
    return _lin(sd, "final_layer.linear", (1 + scale[:, None, :]) * _layer_norm(x) + shift[:, None, :])


# ----------------------------------------------------------------------------- model
def prepare_block_inputs(sd, cfg, img, img_ids, txt, txt_ids, timesteps, y_vec, cond=None, guidance=None):
    """MMDiTModel.prepare_block_inputs, model.py:154-202.  Returns img, txt, vec, angles[B,L,hd/2]."""
    if img.ndim != 3 or txt.ndim != 3:
        raise ValueError("Input img and txt tensors must have 3 dimensions.")
    img = _lin(sd, "img_in", img)
    if cfg.get("cond_embed", False):
        if cond is None:
            raise ValueError("Didn't get conditional input for conditional model.")
        img = img + _lin(sd, "cond_in", cond)
    vec = mlp_embedder(sd, "time_in", timestep_embedding(timesteps, 256))
    if cfg.get("guidance_embed", False):
        if guidance is None:
            raise ValueError("Didn't get guidance strength for guidance distilled model.")
        vec = vec + mlp_embedder(sd, "guidance_in", timestep_embedding(guidance, 256))
    vec = vec + mlp_embedder(sd, "vector_in", y_vec)
    txt = _lin(sd, "txt_in", txt)
    ids = torch.cat((txt_ids, img_ids), dim=1)
    if cfg.get("use_liger_rope", False):
        ang = rope_angles_liger(ids, cfg["axes_dim"], cfg["theta"])
    else:
        ang = rope_angles(ids, cfg["axes_dim"], cfg["theta"])
    return img, txt, vec, ang


def forward(sd, cfg, img, img_ids, txt, txt_ids, timesteps, y_vec, cond=None, guidance=None, taps: dict | None = None):
    """MMDiTModel.forward_ckpt, model.py:208-233.  `taps`, if given, receives intermediates."""
    rope_mode = "half" if cfg.get("use_liger_rope", False) else "interleaved"
    img, txt, vec, ang = prepare_block_inputs(sd, cfg, img, img_ids, txt, txt_ids, timesteps, y_vec, cond, guidance)
    if taps is not None:
        taps.update(img_in=img, txt_in=txt, vec=vec, ang=ang)
    for i in range(cfg["depth"]):
        img, txt = double_block(sd, cfg, i, img, txt, vec, ang, rope_mode)
        if taps is not None:
            taps[f"double.{i}.img"] = img
            taps[f"double.{i}.txt"] = txt
    x = torch.cat((txt, img), 1)
    for i in range(cfg["depth_single_blocks"]):
        x = single_block(sd, cfg, i, x, vec, ang, rope_mode)
        if taps is not None:
            taps[f"single.{i}"] = x
    x = x[:, txt.shape[1] :]
    return last_layer(sd, x, vec)


def flops_per_forward(cfg: dict, B: int, L_img: int, L_txt: int) -> float:
    """SURVEY.md §8(d) algorithmic FLOPs per forward (multiply-add = 2, attention not halved)."""
    D = cfg["hidden_size"]
    r = cfg["mlp_ratio"]
    nd, ns = cfg["depth"], cfg["depth_single_blocks"]
    L = L_img + L_txt
    C = cfg["in_channels"]
    p2 = cfg.get("patch_size", 2) ** 2
    f = (nd + ns) * ((8 + 4 * r) * B * L * D * D + 4 * B * L * L * D)
    f += 2 * B * L_img * D * (C + (C + p2 if cfg.get("cond_embed") else 0) + C)
    f += 2 * B * L_txt * cfg["context_in_dim"] * D
    f += 2 * B * D * D * (12 * nd + 3 * ns + 2)
    f += 2 * B * (256 + cfg["vec_in_dim"] + 2 * D) * D
    return float(f)
