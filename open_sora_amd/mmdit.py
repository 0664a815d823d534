"""MI355X-native MMDiT denoiser behind the reference's module API.

Mirrors, by name and call signature (state-dict keys unchanged, so reference checkpoints load):
    MMDiTConfig / MMDiTModel / Flux()        /root/reference/opensora/models/mmdit/model.py:40-303
    DoubleStreamBlock / SingleStreamBlock     /root/reference/opensora/models/mmdit/layers.py:256-306,337-388
    set_processor()/get_processor() plug-in   layers.py:299-303,381-385
    MLPEmbedder, Modulation, QKNorm, SelfAttention, LastLayer   layers.py:91-192,391-402

The nn.Modules here only HOLD parameters.  All arithmetic runs in hand-written gfx950 kernels reached
through the C ABI of include/osk.h (open_sora_amd/_C.py); there is no eager/PyTorch compute fallback —
if libosk_hip.so is missing the import of this module fails.

Two entry levels, same kernels:
  * MMDiTModel.forward(...)  — whole-step engine: one batched adaLN GEMV for all 57 blocks, activations
    kept in pre-allocated joint [txt;img] buffers so no torch.cat / rearrange copy is ever materialised.
  * HipDoubleStreamBlockProcessor / HipSingleStreamBlockProcessor — callables with the reference
    processor signature `(block, img, txt, vec, pe)` / `(block, x, vec, pe)`; they read weights from the
    block's submodules by the reference's attribute names, so they can be installed with
    `block.set_processor(...)` on the reference's own DoubleStreamBlock / SingleStreamBlock as well.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import weakref

import torch
from torch import Tensor, nn

from . import _C

# Kernel table used by the orchestration below.  The product binds it to the HIP library (_C) at import —
# there is no other implementation in this package.  tests/ may swap in a CPU emulation of the kernels'
# semantics (tests/cpu_ops.py) to exercise the host-side orchestration and the multi-process sequence-parallel
# path under gloo without a GPU; that emulation is checker infrastructure and never ships.
_OPS = _C


def set_ops_for_testing(ops) -> None:
    global _OPS
    _OPS = ops     # workspaces are keyed on id(_OPS): buffers made for another kernel table are not reused


def ops():
    return _OPS

BF16 = torch.bfloat16


# =============================================================================================
# parameter containers (names == reference state-dict keys)
# =============================================================================================
@dataclass
class MMDiTConfig:
    """Field-for-field the reference MMDiTConfig (model.py:40-67)."""

    model_type = "MMDiT"
    from_pretrained: str | None
    cache_dir: str | None
    in_channels: int
    vec_in_dim: int
    context_in_dim: int
    hidden_size: int
    mlp_ratio: float
    num_heads: int
    depth: int
    depth_single_blocks: int
    axes_dim: list
    theta: int
    qkv_bias: bool
    guidance_embed: bool
    cond_embed: bool = False
    fused_qkv: bool = True
    grad_ckpt_settings: tuple | None = None
    use_liger_rope: bool = False
    patch_size: int = 2

    def get(self, attribute_name, default=None):
        return getattr(self, attribute_name, default)

    def __contains__(self, attribute_name):
        return hasattr(self, attribute_name)


class _OskState:
    """mix-in: the kernel-side caches hung on a module (`_osk_plan`: derived weight images and raw device pointers,
    `_osk_ws_cache`: activation workspaces) are rebuilt on demand and must not travel with copy.deepcopy / pickle / torch.save
    (they would duplicate or serialise large device buffers: ADVICE r2)."""

    def __getstate__(self):
        d = self.__dict__.copy()
        for k in [k for k in d if k.startswith("_osk_")]:
            del d[k]
        if "_plan" in d:
            d["_plan"] = None
        return d


# load_state_dict watchers.  The hook itself is a module-level function (module hooks are pickled / deep-copied with the module:
# a closure would break torch.save(model)); who it serves lives in weak side tables that never travel.
_WATCHERS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()   # sub-module -> [weakref(owner of a cached plan)]
_WATCHED: "weakref.WeakSet" = weakref.WeakSet()                        # owners already wired


def _state_loaded(module, _incompatible_keys) -> None:
    for ref in _WATCHERS.get(module, ()):
        b = ref()
        if b is None:
            continue
        if hasattr(b, "invalidate_plan"):        # the whole model: its plan also holds f32 copies of the blocks' adaLN biases
            b.invalidate_plan()
        elif "_osk_plan" in b.__dict__:
            object.__delattr__(b, "_osk_plan")


def _watch_state_loads(owner) -> None:
    """A load_state_dict at any level of the owner's module tree (the block, one of its Linear layers, ...) drops the owner's
    cached plan -- also under torch.inference_mode(), where the (pointer, version) key of _param_key cannot see an in-place
    copy.  Works on the reference's own blocks too (the processors cache their plans on whatever block they run on)."""
    if owner in _WATCHED:
        return
    for m in owner.modules():
        if _state_loaded not in m._load_state_dict_post_hooks.values():
            m.register_load_state_dict_post_hook(_state_loaded)
        _WATCHERS.setdefault(m, []).append(weakref.ref(owner))
    _WATCHED.add(owner)


class _Holder(nn.Module):
    """A module that owns parameters but whose arithmetic lives in the HIP engine."""

    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError(
            f"{type(self).__name__} holds parameters only; its arithmetic runs in libosk_hip.so via the "
            "block processor / MMDiTModel.forward (no eager fallback)."
        )


class MLPEmbedder(_Holder):
    def __init__(self, in_dim: int, hidden_dim: int):
        super().__init__()
        self.in_layer = nn.Linear(in_dim, hidden_dim, bias=True)
        self.silu = nn.SiLU()
        self.out_layer = nn.Linear(hidden_dim, hidden_dim, bias=True)


class RMSNorm(_Holder):
    def __init__(self, dim: int):
        super().__init__()
        self.scale = nn.Parameter(torch.ones(dim))


class QKNorm(_Holder):
    def __init__(self, dim: int):
        super().__init__()
        self.query_norm = RMSNorm(dim)
        self.key_norm = RMSNorm(dim)


class SelfAttention(_Holder):
    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = False, fused_qkv: bool = True):
        super().__init__()
        self.num_heads = num_heads
        self.fused_qkv = fused_qkv
        if fused_qkv:
            self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        else:
            self.q_proj = nn.Linear(dim, dim, bias=qkv_bias)
            self.k_proj = nn.Linear(dim, dim, bias=qkv_bias)
            self.v_proj = nn.Linear(dim, dim, bias=qkv_bias)
        self.norm = QKNorm(dim // num_heads)
        self.proj = nn.Linear(dim, dim)


class Modulation(_Holder):
    def __init__(self, dim: int, double: bool):
        super().__init__()
        self.is_double = double
        self.multiplier = 6 if double else 3
        self.lin = nn.Linear(dim, self.multiplier * dim, bias=True)


class _NoParamNorm(_Holder):
    """Stands for nn.LayerNorm(D, elementwise_affine=False, eps=1e-6): no parameters, fused into osk_ln_modulate."""

    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.normalized_shape = (dim,)
        self.eps = eps


class DoubleStreamBlock(_OskState, nn.Module):
    def __init__(self, hidden_size: int, num_heads: int, mlp_ratio: float, qkv_bias: bool = False,
                 fused_qkv: bool = True):
        super().__init__()
        mlp_hidden = int(hidden_size * mlp_ratio)
        self.num_heads = num_heads
        self.hidden_size = hidden_size
        self.head_dim = hidden_size // num_heads
        for s in ("img", "txt"):
            setattr(self, f"{s}_mod", Modulation(hidden_size, double=True))
            setattr(self, f"{s}_norm1", _NoParamNorm(hidden_size))
            setattr(self, f"{s}_attn", SelfAttention(hidden_size, num_heads, qkv_bias, fused_qkv))
            setattr(self, f"{s}_norm2", _NoParamNorm(hidden_size))
            setattr(self, f"{s}_mlp", nn.Sequential(
                nn.Linear(hidden_size, mlp_hidden, bias=True),
                nn.GELU(approximate="tanh"),
                nn.Linear(mlp_hidden, hidden_size, bias=True),
            ))
        self.set_processor(HipDoubleStreamBlockProcessor())

    def set_processor(self, processor) -> None:
        self.processor = processor

    def get_processor(self):
        return self.processor

    def forward(self, img: Tensor, txt: Tensor, vec: Tensor, pe, **kwargs):
        return self.processor(self, img, txt, vec, pe)


class SingleStreamBlock(_OskState, nn.Module):
    def __init__(self, hidden_size: int, num_heads: int, mlp_ratio: float = 4.0, qk_scale: float | None = None,
                 fused_qkv: bool = True):
        super().__init__()
        self.hidden_dim = hidden_size
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.head_dim = hidden_size // num_heads
        self.scale = qk_scale or self.head_dim ** -0.5
        self.fused_qkv = fused_qkv
        self.mlp_hidden_dim = int(hidden_size * mlp_ratio)
        if fused_qkv:
            self.linear1 = nn.Linear(hidden_size, hidden_size * 3 + self.mlp_hidden_dim)
        else:
            self.q_proj = nn.Linear(hidden_size, hidden_size)
            self.k_proj = nn.Linear(hidden_size, hidden_size)
            self.v_mlp = nn.Linear(hidden_size, hidden_size + self.mlp_hidden_dim)
        self.linear2 = nn.Linear(hidden_size + self.mlp_hidden_dim, hidden_size)
        self.norm = QKNorm(self.head_dim)
        self.pre_norm = _NoParamNorm(hidden_size)
        self.mlp_act = nn.GELU(approximate="tanh")
        self.modulation = Modulation(hidden_size, double=False)
        self.set_processor(HipSingleStreamBlockProcessor())

    def set_processor(self, processor) -> None:
        self.processor = processor

    def get_processor(self):
        return self.processor

    def forward(self, x: Tensor, vec: Tensor, pe, **kwargs) -> Tensor:
        return self.processor(self, x, vec, pe)


class LastLayer(_Holder):
    def __init__(self, hidden_size: int, patch_size: int, out_channels: int):
        super().__init__()
        self.norm_final = _NoParamNorm(hidden_size)
        self.linear = nn.Linear(hidden_size, patch_size * patch_size * out_channels, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 2 * hidden_size, bias=True))


# =============================================================================================
# weight preparation helpers
# =============================================================================================
def _w(t: Tensor) -> Tensor:
    t = t.detach()
    return t if (t.dtype == BF16 and t.is_contiguous()) else t.to(BF16).contiguous()


def _b32(t: Tensor | None) -> Tensor | None:
    return None if t is None else t.detach().float().contiguous()


def _cat_linear(*lins):
    w = torch.cat([_w(l.weight) for l in lins], 0).contiguous() if len(lins) > 1 else _w(lins[0].weight)
    if lins[0].bias is None:
        return w, None
    b = torch.cat([l.bias.detach().float() for l in lins], 0).contiguous()
    return w, b


def _pad_k(w: Tensor, mult: int = 64) -> Tensor:
    N, K = w.shape
    Kp = (K + mult - 1) // mult * mult
    if Kp == K:
        return w
    out = torch.zeros(N, Kp, dtype=w.dtype, device=w.device)
    out[:, :K] = w
    return out


class Fp8Weight:
    """A Linear weight in both precisions for the opt-in fp8 mode (MMDiTModel.enable_fp8): OCP e4m3 bytes with one
    f32 scale per output row for the large-tile fp8 GEMM, and the bf16 rows for the shapes that kernel does not take
    (M < 256: e.g. a short text shard under sequence parallelism).  Row slices stay views."""

    def __init__(self, w: Tensor, w8: Tensor | None = None, sw: Tensor | None = None):
        self.w = w
        if w8 is None:
            w8, sw = _OPS.quantize_rows_fp8(w)
        self.w8, self.sw = w8, sw

    @property
    def shape(self):
        return self.w.shape

    def __getitem__(self, rows):
        assert isinstance(rows, slice) and (rows.start or 0) % 4 == 0   # the scale vector is read 16 bytes at a time
        return Fp8Weight(self.w[rows], self.w8[rows], self.sw[rows])


def _linear(a, w, b, out: Tensor, **epi) -> Tensor:
    """nn.Linear + fused epilogue: bf16 GEMM, or -- when w is an Fp8Weight and the shape qualifies -- dynamic per-row
    quantisation of the activation followed by the fp8 GEMM.  `a` is a bf16 [B, L, K] view or an already quantised
    activation (a8, row scales) from _ln_modulate_for()."""
    if isinstance(a, tuple):
        a8, sa = a
        return _OPS.gemm_fp8(a8, sa, w.w8, w.sw, b, out, **epi)
    if isinstance(w, Fp8Weight):
        B, L, K = a.shape
        if B * L > 0 and _OPS.gemm_fp8_supported(B * L, w.shape[0], K):
            a8, sa = _OPS.quantize_rows_fp8(a)
            return _OPS.gemm_fp8(a8, sa, w.w8, w.sw, b, out, **epi)
        w = w.w
    return _OPS.gemm(a, w, b, out, **epi)


def _linear_pair(first: dict, second: dict, gelu_from: int | None = None) -> None:
    """the same Linear kind on the img and on the txt stream (dicts of _linear's arguments a, w, bias, out [, res, gate,
    gate_batch_stride]): one launch (osk_gemm_bf16_pair: the text GEMM fills the image GEMM's last round of tiles) for bf16
    operands, the two single calls in fp8 mode"""
    if all(not isinstance(d["a"], tuple) and not isinstance(d["w"], Fp8Weight) for d in (first, second)):
        _OPS.gemm_pair(first, second, gelu_from=gelu_from)
        return
    for d in (first, second):
        epi = {k: d[k] for k in ("res", "gate", "gate_batch_stride") if k in d}
        if gelu_from is not None:
            epi["gelu_from"] = gelu_from
        _linear(d["a"], d["w"], d["bias"], d["out"], **epi)


def _ln_modulate_for(w, x: Tensor, shift: Tensor, scale: Tensor, xm: Tensor, mbs: int):
    """LN + modulate of x as the input of Linear layer(s) with weight w (row slices of w included): the bf16 rows in
    xm, or -- fp8 mode, shapes the fp8 GEMM takes -- quantised on the fly (one pass, nothing written to xm)."""
    B, L, K = x.shape
    if isinstance(w, Fp8Weight) and B * L > 0 and _OPS.gemm_fp8_supported(B * L, 128, K):
        return _OPS.ln_modulate_fp8(x, shift, scale, mbs)
    _OPS.ln_modulate(x, shift, scale, xm, mbs)
    return xm


@dataclass
class _AttnW:
    qkv_w: Tensor
    qkv_b: Tensor | None
    q_scale: Tensor
    k_scale: Tensor
    proj_w: Tensor | None = None
    proj_b: Tensor | None = None


@dataclass
class _DoublePlan:
    img: _AttnW
    txt: _AttnW
    img_mlp: tuple = ()
    txt_mlp: tuple = ()
    mod_layers: list = field(default_factory=list)  # [(w bf16, b bf16)] img then txt
    pv8: bool = False   # fp8 mode: attention with the P.V product on the fp8 MFMA
    key: tuple = ()     # _param_key of the block the plan was built from
    score_bound: float = 0.0   # bound on |q . k| as the attention kernel sees it (log2 units): _score_bound()


@dataclass
class _SinglePlan:
    w1: Tensor
    b1: Tensor | None
    w2: Tensor
    b2: Tensor | None
    q_scale: Tensor
    k_scale: Tensor
    mod_layers: list = field(default_factory=list)
    pv8: bool = False
    key: tuple = ()
    score_bound: float = 0.0


def _score_bound(hd: int, q_scales, k_scales) -> float:
    """Upper bound on |q . k| for the q, k the attention kernel receives from osk_qknorm_rope_bf16 (log2 units: q carries
    q_mult(hd) = hd^-1/2 log2 e).  QKNorm (layers.py:102-135) = RMS norm times a learned scale vector: sum_i (x_i rrms)^2 <= hd,
    so |q| <= sqrt(hd) max|w_q| and |k| <= sqrt(hd) max|w_k|; RoPE is a rotation; the three bf16 roundings on each side add
    < 2.5 % (covered by the 1.05).  Cauchy-Schwarz gives the bound the fast attention body uses as its softmax reference
    (include/osk.h, osk_attention_fwd_bounded_bf16)."""
    wq = max(float(t.detach().float().abs().max()) for t in q_scales)
    wk = max(float(t.detach().float().abs().max()) for t in k_scales)
    return 1.05 * q_mult(hd) * hd * wq * wk


def _attn_weights(sa, wrap=lambda w: w) -> _AttnW:
    if getattr(sa, "fused_qkv", hasattr(sa, "qkv")):
        w, b = _cat_linear(sa.qkv)
    else:
        w, b = _cat_linear(sa.q_proj, sa.k_proj, sa.v_proj)
    return _AttnW(wrap(w), b, _w(sa.norm.query_norm.scale), _w(sa.norm.key_norm.scale), wrap(_w(sa.proj.weight)),
                  _b32(sa.proj.bias))


def _wrap_for(block):
    """weights of a block whose model runs in fp8 mode are kept as Fp8Weight (quantised once, here)"""
    return Fp8Weight if getattr(block, "_osk_fp8", False) else (lambda w: w)


def _mod_layer(mod) -> tuple:
    return (_w(mod.lin.weight), None if mod.lin.bias is None else _w(mod.lin.bias))


def _tensor_key(p: Tensor) -> tuple:
    """(storage pointer, in-place version counter) of one parameter.  Inference tensors (parameters created or loaded
    under torch.inference_mode(), as the reference's inference scripts do) do not track a version counter -- reading
    `_version` raises on them -- and cannot be written in place outside inference mode either: their version is 0."""
    return (p.data_ptr(), 0 if p.is_inference() else p._version)


def _param_key(module):
    """_tensor_key of every parameter of the module: a cached plan built from other values is stale.  Detected: a storage
    swap (load_state_dict, .to(), a new Parameter) and autograd-visible in-place writes (`p.mul_()`, `p.copy_()` under
    no_grad, an optimizer step).  NOT detected: writes through `p.data` (`.data` is a detached alias with its own version
    counter) and writes to inference tensors -- after those, call `model.invalidate_plan()`."""
    return tuple(_tensor_key(p) for p in module.parameters())


def plan_double(block) -> _DoublePlan:
    p = getattr(block, "_osk_plan", None)
    if p is not None and p.key != _param_key(block):
        p = None
    if p is None:
        wr = _wrap_for(block)
        p = _DoublePlan(
            img=_attn_weights(block.img_attn, wr),
            txt=_attn_weights(block.txt_attn, wr),
            img_mlp=(wr(_w(block.img_mlp[0].weight)), _b32(block.img_mlp[0].bias), wr(_w(block.img_mlp[2].weight)), _b32(block.img_mlp[2].bias)),
            txt_mlp=(wr(_w(block.txt_mlp[0].weight)), _b32(block.txt_mlp[0].bias), wr(_w(block.txt_mlp[2].weight)), _b32(block.txt_mlp[2].bias)),
            mod_layers=[_mod_layer(block.img_mod), _mod_layer(block.txt_mod)],
            pv8=bool(getattr(block, "_osk_fp8", False)),
        )
        hd = block.head_dim if hasattr(block, "head_dim") else block.hidden_size // block.num_heads
        p.score_bound = _score_bound(hd, (p.img.q_scale, p.txt.q_scale), (p.img.k_scale, p.txt.k_scale))
        p.key = _param_key(block)
        _watch_state_loads(block)
        object.__setattr__(block, "_osk_plan", p)
    return p


def plan_single(block) -> _SinglePlan:
    p = getattr(block, "_osk_plan", None)
    if p is not None and p.key != _param_key(block):
        p = None
    if p is None:
        if getattr(block, "fused_qkv", hasattr(block, "linear1")):
            w1, b1 = _cat_linear(block.linear1)
        else:
            w1, b1 = _cat_linear(block.q_proj, block.k_proj, block.v_mlp)
        wr = _wrap_for(block)
        p = _SinglePlan(wr(w1), b1, wr(_w(block.linear2.weight)), _b32(block.linear2.bias),
                        _w(block.norm.query_norm.scale), _w(block.norm.key_norm.scale),
                        mod_layers=[_mod_layer(block.modulation)], pv8=bool(getattr(block, "_osk_fp8", False)))
        hd = block.head_dim if hasattr(block, "head_dim") else block.hidden_size // block.num_heads
        p.score_bound = _score_bound(hd, (p.q_scale,), (p.k_scale,))
        p.key = _param_key(block)
        _watch_state_loads(block)
        object.__setattr__(block, "_osk_plan", p)
    return p


# =============================================================================================
# workspace: activations for one (B, L_img, L_txt) geometry, allocated once and reused every step
# =============================================================================================
class _Workspace:
    def __init__(self, B, L_txt, L_img, D, R, H, hd, device):
        L = L_txt + L_img
        self.B, self.L_txt, self.L_img, self.L = B, L_txt, L_img, L
        self.x = torch.empty(B, L, D, dtype=BF16, device=device)        # residual stream, joint [txt; img]
        self.xm = torch.empty(B, L, D, dtype=BF16, device=device)       # LN+modulate output
        self.y = torch.empty(B * L * (3 * D + R), dtype=BF16, device=device)  # [q|k|v(attn out)|mlp] rows
        self.h = torch.empty(B, L, R, dtype=BF16, device=device)        # double-block MLP hidden
        Lp = (L + 63) // 64 * 64
        self.vt = torch.empty(B, H, hd, Lp, dtype=BF16, device=device)
        # the QKV projection writes V directly as V^T (osk_gemm_group_bf16) until the library declines a group of this geometry
        self.vt_group = hasattr(_OPS, "gemm_group")

    def y_double(self, D):
        return self.y[: self.B * self.L * 3 * D].view(self.B, self.L, 3 * D)

    def y_single(self, D, R):
        return self.y.view(self.B, self.L, 3 * D + R)


def _stream_key(device) -> int:
    """identity of the stream the kernels of this call are enqueued on (0 on the CPU emulation used by tests)"""
    return torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0


def _workspace(owner, B, L_txt, L_img, D, R, H, hd, device) -> _Workspace:
    """Activation buffers of one (geometry, device, STREAM), owned by `owner` (the model, or the block a stand-alone
    processor call runs on): two models, or one model driven from two streams, never share buffers -- launches are only
    ordered within a stream.  Allocated once per key and reused by every block and step; at most 4 geometries are kept."""
    cache = getattr(owner, "_osk_ws_cache", None)
    if cache is None:
        cache = {}
        object.__setattr__(owner, "_osk_ws_cache", cache)
    key = (B, L_txt, L_img, D, R, H, hd, str(device), _stream_key(device), id(_OPS))
    ws = cache.pop(key, None)
    if ws is None:
        while len(cache) >= 4:   # least recently used first (dicts keep insertion order; a hit is re-inserted below).  The entry in
            cache.pop(next(iter(cache)))   # use -- e.g. the one a captured hipGraph replays into -- is always the newest
        ws = _Workspace(B, L_txt, L_img, D, R, H, hd, device)
    cache[key] = ws
    return ws


# =============================================================================================
# RoPE table handling
# =============================================================================================
_PE_MEMO: list = [None]   # (weakrefs of the pe tensor(s), versions, hd, device, stream key, result): the conversion of the LAST pe seen


def _pe_to_cos_sin(pe, hd: int):
    """Reference positional-embedding formats -> (cos, sin) f32 [B, L, hd/2], rope_mode.
    EmbedND tensor [B,1,L,hd/2,2,2] (layers.py:38-44; entries cos,-sin,sin,cos) -> mode 0 (interleaved);
    LigerEmbedND tuple of [B,L,hd] (layers.py:55-65; halves repeated)           -> mode 1 (half-split).
    A reference model hands the SAME pe object to each of its blocks (model.py:218-229): the conversion (two strided
    gathers) is done once per object and remembered while that object is alive and unmodified -- 57 processor calls of a
    forward pay for one."""
    if isinstance(pe, _RopeTable):
        return pe.cos, pe.sin, pe.mode
    parts = (pe,) if isinstance(pe, torch.Tensor) else tuple(pe)
    # The memo: keyed on the tensors' identity AND version counter, the device and the STREAM the conversion was
    # enqueued on (a result produced on one stream is not ordered before work on another); held through weak references (it must
    # not extend the life of a caller's pe); bypassed for inference tensors (they keep no version counter but CAN be refilled in
    # place under torch.inference_mode()) and while the stream is being captured into a hipGraph (a result produced inside a
    # capture only exists when the graph is replayed).
    dev = parts[0].device
    memo_ok = not any(t.is_inference() for t in parts) and not (dev.type == "cuda" and torch.cuda.is_current_stream_capturing())
    vers = tuple(t._version for t in parts) if memo_ok else ()
    skey = _stream_key(dev)
    m = _PE_MEMO[0]
    if (memo_ok and m is not None and m[2] == hd and m[3] == str(dev) and m[4] == skey and len(m[0]) == len(parts)
            and all(r() is b for r, b in zip(m[0], parts)) and m[1] == vers):
        return m[5]
    if isinstance(pe, torch.Tensor):
        res = (pe[:, 0, :, :, 0, 0].float().contiguous(), pe[:, 0, :, :, 1, 0].float().contiguous(), 0)
    else:
        cos, sin = pe
        res = (cos[..., : hd // 2].float().contiguous(), sin[..., : hd // 2].float().contiguous(), 1)
    if memo_ok:
        _PE_MEMO[0] = (tuple(weakref.ref(t) for t in parts), vers, hd, str(dev), skey, res)
    return res


@dataclass
class _RopeTable:
    cos: Tensor  # f32 [B, L, hd/2]
    sin: Tensor
    mode: int    # 0 interleaved (EmbedND), 1 half-split (LigerEmbedND)


class HipEmbedND(nn.Module):
    """EmbedND / LigerEmbedND (layers.py:31-65): ids [B, L, n_axes] -> RoPE tables, computed by osk_rope_table."""

    def __init__(self, dim: int, theta: int, axes_dim, liger: bool):
        super().__init__()
        self.dim, self.theta, self.axes_dim, self.liger = dim, theta, list(axes_dim), liger

    def forward(self, ids: Tensor) -> _RopeTable:
        B, L, n_axes = ids.shape
        idf = ids.float().contiguous().view(B * L, n_axes)
        half = sum(self.axes_dim) // 2
        cos = torch.empty(B, L, half, dtype=torch.float32, device=ids.device)
        sin = torch.empty_like(cos)
        _OPS.rope_table(idf, self.axes_dim, self.theta, self.liger, cos, sin)
        return _RopeTable(cos, sin, 1 if self.liger else 0)


# =============================================================================================
# the block arithmetic (shared by the engine and the processors)
# =============================================================================================
def _mod_views(mod: Tensor, col: int, n: int, D: int):
    """mod f32 [B, Ntot]; returns n views (pointer carriers) at columns col + i*D, and the batch stride."""
    return [mod[:, col + i * D: col + (i + 1) * D] for i in range(n)], mod.stride(0)


# blocks whose weight-derived score bound exceeds the FAST body's limit take the bound from their operands on the device
# (osk_attention_fwd_auto_bf16) instead of falling to the general body wholesale; False = round 5's behaviour
AUTO_BOUND = True
SCORE_BOUND_LIMIT = 56.0   # csrc/attention_params.h: OSK_ATTN_MAX_BOUND


def q_mult(hd: int) -> float:
    """softmax scale * log2(e): folded into q by the QK-norm + RoPE kernel before q's single rounding to bf16, so the
    attention kernels exponentiate (in base 2) what the MFMA hands them (include/osk.h, q_prescaled)."""
    return hd ** -0.5 * 1.4426950408889634


def v_scale_fp8(v: Tensor, H: int, hd: int) -> Tensor:
    """one e4m3 scale per (batch, head) of a [B, L, H*hd] V view: absmax / 448 (f32 [B, H], device)"""
    return _OPS.v_scale_fp8(v, H, hd)


def _bf16_operands(*xs) -> bool:
    """plain bf16 activations / weights (not the fp8 mode's quantised pairs)"""
    return all(not isinstance(x, (tuple, Fp8Weight)) for x in xs)


def _joint_attention(ws: _Workspace, q: Tensor, k: Tensor, v: Tensor, H: int, hd: int, pv8: bool = False,
                     score_bound: float = 0.0, vt_ready: bool = False):
    """attention() of math.py:22-36 on the joint [txt;img] sequence; the output overwrites the (dead) v slot.
    pv8 (fp8 mode, head_dim 72 / 128): V^T as e4m3 with one scale per (batch, head), P.V on the fp8 MFMA.
    vt_ready: ws.vt was already written by the projection (osk_gemm_group_bf16's V^T task) -- v then only names the output slot."""
    wsp = _OPS.attention_workspace(q.device)
    if pv8 and hd in (72, 128):
        B = v.shape[0]
        vt8 = getattr(ws, "vt8", None)
        if vt8 is None:
            vt8 = ws.vt8 = torch.empty(B, H, _OPS.vt8_rows(hd), ws.vt.shape[-1], dtype=torch.uint8, device=v.device)
        sv = v_scale_fp8(v, H, hd)
        _OPS.v_transpose_fp8(v, sv, vt8, H, hd)
        _OPS.attention_fwd_pv8(q, k, vt8, sv, v, H, hd, hd ** -0.5, q_prescaled=True, workspace=wsp)
        return
    if not vt_ready:
        _OPS.v_transpose(v, ws.vt, H, hd)
    if AUTO_BOUND and not (0.0 < score_bound <= SCORE_BOUND_LIMIT) and hasattr(_OPS, "attention_fwd_auto"):
        # The weight-derived bound does not admit the FAST body (a checkpoint whose QK-norm scale vectors have large entries: the
        # bound is sqrt(hd)-loose per side).  Take the bound from the operands instead, on the device: squared row-norm maxima of
        # this call's q and k per (batch, head), then the auto-dispatched launch pair -- every (batch, head) whose actual
        # |q| |k| <= 56 runs the FAST body, the rest the general one; no host round trip (include/osk.h).
        n2 = getattr(ws, "qk_n2", None)
        if n2 is None:
            n2 = ws.qk_n2 = torch.empty(2, q.shape[0], H, dtype=torch.float32, device=q.device)
        _OPS.rownorm2_max(q, n2[0], H, hd)
        _OPS.rownorm2_max(k, n2[1], H, hd)
        _OPS.attention_fwd_auto(q, k, ws.vt, v, H, hd, hd ** -0.5, n2[0], n2[1], q_prescaled=True, workspace=wsp)
        return
    _OPS.attention_fwd(q, k, ws.vt, v, H, hd, hd ** -0.5, q_prescaled=True, workspace=wsp, score_bound=score_bound)


def run_double_block(plan: _DoublePlan, ws: _Workspace, mod: Tensor, col_img: int, col_txt: int, rope: _RopeTable,
                     H: int, hd: int, sp=None, x_in=None, x_out=None):
    """DoubleStreamBlockProcessor.__call__ (layers.py:195-253) on the workspace's joint buffers.
    Residual streams live in ws.x ([:, :L_txt] txt, [:, L_txt:] img) and are updated in place.
    sp: a seqpar.SeqPar when the token axis is sharded (ws then holds this rank's rows; either stream may be
    empty on a rank) — K/V are projected first and all-gathered while the Q projection runs.
    x_in / x_out = (img, txt) pairs (stand-alone processor calls, both streams present, sp None): the block READS its input
    streams from x_in (first LayerNorm, residual of the attention projection) and continues the residual streams in x_out, so a
    caller's tensors are neither copied in nor modified."""
    D = H * hd
    Lt, Li = ws.L_txt, ws.L_img
    x_txt, x_img = ws.x[:, :Lt], ws.x[:, Lt:]
    r_img, r_txt = x_img, x_txt            # where the block's input streams are read from
    if x_in is not None:
        assert sp is None and Li and Lt and x_out is not None
        (r_img, r_txt), (x_img, x_txt) = x_in, x_out
    xm_txt, xm_img = ws.xm[:, :Lt], ws.xm[:, Lt:]
    y = ws.y_double(D)
    (i_sh1, i_sc1, i_g1, i_sh2, i_sc2, i_g2), mbs = _mod_views(mod, col_img, 6, D)
    (t_sh1, t_sc1, t_g1, t_sh2, t_sc2, t_g2), _ = _mod_views(mod, col_txt, 6, D)
    csb = rope.cos.stride(0) if rope.cos.shape[0] > 1 else 0
    streams = []
    if Li:
        streams.append((plan.img, r_img, xm_img, y[:, Lt:], i_sh1, i_sc1))
    if Lt:
        streams.append((plan.txt, r_txt, xm_txt, y[:, :Lt], t_sh1, t_sc1))

    acts = [_ln_modulate_for(aw.qkv_w, x_s, sh1, sc1, xm_s, mbs) for aw, x_s, xm_s, y_s, sh1, sc1 in streams]
    q, k, v = y[:, :, :D], y[:, :, D: 2 * D], y[:, :, 2 * D:]
    scales = (plan.txt.q_scale, plan.txt.k_scale, plan.img.q_scale, plan.img.k_scale)
    paired = sp is None and len(streams) == 2      # both streams on this rank: their Linear layers go out in pairs
    vt_ready = False
    if (paired and ws.vt_group and not plan.pv8 and Lt % 64 == 0 and _bf16_operands(*acts, plan.img.qkv_w, plan.txt.qkv_w)):
        # ONE launch for the block's four projection problems: q | k of each stream into the row buffer, V of each stream straight
        # into ws.vt as the key-major operand of the attention kernel (txt keys first) -- no V round trip, no osk_v_transpose_bf16
        tasks = []
        for ((aw, x_s, xm_s, y_s, sh1, sc1), act), pos in zip(zip(streams, acts), (Lt, 0)):     # streams = [img, txt]
            b = aw.qkv_b
            tasks.append(dict(a=act, w=aw.qkv_w[:2 * D], bias=None if b is None else b[:2 * D], out=y_s))
            tasks.append(dict(x=act, w=aw.qkv_w[2 * D:], bias=None if b is None else b[2 * D:], vt=ws.vt, vt_pos=pos, hd=hd))
        vt_ready = _OPS.gemm_group(tasks)
        if not vt_ready:
            ws.vt_group = False                    # (a shape off the 256 x 256 tile path: the single calls from now on)
    if vt_ready:
        pass
    elif paired:
        _linear_pair(*(dict(a=act, w=aw.qkv_w, bias=aw.qkv_b, out=y_s) for (aw, x_s, xm_s, y_s, sh1, sc1), act in zip(streams, acts)))
    elif sp is None:
        for (aw, x_s, xm_s, y_s, sh1, sc1), act in zip(streams, acts):
            _linear(act, aw.qkv_w, aw.qkv_b, y_s)
    if sp is None:
        _OPS.qknorm_rope(q, k, *scales, Lt, rope.cos, rope.sin, csb, H, hd, rope.mode, q_mult=q_mult(hd))
        _joint_attention(ws, q, k, v, H, hd, plan.pv8, plan.score_bound, vt_ready)
    else:
        # K, V first: their exchange overlaps the Q projection.  All-gather mode, bf16: V goes straight into this rank's slot of the
        # gathered V^T buffer (osk_gemm_group_bf16's V^T task, round 6) -- no token-major V, no osk_v_transpose_bf16 pass on the rank
        vt_ready = False
        vt_loc = sp.local_vt(ws.B, ws.L, H, hd, ws.x.device) if (ws.vt_group and not plan.pv8 and Lt % 64 == 0 and hasattr(sp, "local_vt") and
                                                                  _bf16_operands(*acts, *(s_[0].qkv_w for s_ in streams))) else None
        if vt_loc is not None:
            tasks = []
            for ((aw, x_s, xm_s, y_s, sh1, sc1), act) in zip(streams, acts):
                pos = Lt if (aw is plan.img and Lt) else 0         # local key order: [txt rows ; img rows]
                b = aw.qkv_b
                tasks.append(dict(a=act, w=aw.qkv_w[D: 2 * D], bias=None if b is None else b[D: 2 * D], out=y_s[:, :, D:]))
                tasks.append(dict(x=act, w=aw.qkv_w[2 * D:], bias=None if b is None else b[2 * D:], vt=vt_loc, vt_pos=pos, hd=hd))
            vt_ready = _OPS.gemm_group(tasks)
            if not vt_ready:
                ws.vt_group = False
        if not vt_ready:
            for (aw, x_s, xm_s, y_s, sh1, sc1), act in zip(streams, acts):
                _linear(act, aw.qkv_w[D:], None if aw.qkv_b is None else aw.qkv_b[D:], y_s[:, :, D:])
        _OPS.qknorm_rope(None, k, *scales, Lt, rope.cos, rope.sin, csb, H, hd, rope.mode)
        pending = sp.gather_kv_start(ws, k, v, H, hd, plan.pv8, vt_ready=vt_ready)
        for (aw, x_s, xm_s, y_s, sh1, sc1), act in zip(streams, acts):
            _linear(act, aw.qkv_w[:D], None if aw.qkv_b is None else aw.qkv_b[:D], y_s[:, :, :D])
        _OPS.qknorm_rope(q, None, *scales, Lt, rope.cos, rope.sin, csb, H, hd, rope.mode, q_mult=q_mult(hd))
        sp.attention(ws, pending, q, v, H, hd, plan.score_bound)
    if paired:   # layers.py:247-252 for both streams, Linear by Linear
        _linear_pair(dict(a=v[:, Lt:], w=plan.img.proj_w, bias=plan.img.proj_b, out=x_img, res=r_img, gate=i_g1, gate_batch_stride=mbs),
                     dict(a=v[:, :Lt], w=plan.txt.proj_w, bias=plan.txt.proj_b, out=x_txt, res=r_txt, gate=t_g1, gate_batch_stride=mbs))
        (iw0, ib0, iw2, ib2), (tw0, tb0, tw2, tb2) = plan.img_mlp, plan.txt_mlp
        a_img = _ln_modulate_for(iw0, x_img, i_sh2, i_sc2, xm_img, mbs)
        a_txt = _ln_modulate_for(tw0, x_txt, t_sh2, t_sc2, xm_txt, mbs)
        _linear_pair(dict(a=a_img, w=iw0, bias=ib0, out=ws.h[:, Lt:]), dict(a=a_txt, w=tw0, bias=tb0, out=ws.h[:, :Lt]), gelu_from=0)
        _linear_pair(dict(a=ws.h[:, Lt:], w=iw2, bias=ib2, out=x_img, res=x_img, gate=i_g2, gate_batch_stride=mbs),
                     dict(a=ws.h[:, :Lt], w=tw2, bias=tb2, out=x_txt, res=x_txt, gate=t_g2, gate_batch_stride=mbs))
        return
    if Li:  # img stream
        _linear(v[:, Lt:], plan.img.proj_w, plan.img.proj_b, x_img, res=x_img, gate=i_g1, gate_batch_stride=mbs)
        w0, b0, w2, b2 = plan.img_mlp
        _linear(_ln_modulate_for(w0, x_img, i_sh2, i_sc2, xm_img, mbs), w0, b0, ws.h[:, Lt:], gelu_from=0)
        _linear(ws.h[:, Lt:], w2, b2, x_img, res=x_img, gate=i_g2, gate_batch_stride=mbs)
    if Lt:  # txt stream
        _linear(v[:, :Lt], plan.txt.proj_w, plan.txt.proj_b, x_txt, res=x_txt, gate=t_g1, gate_batch_stride=mbs)
        w0, b0, w2, b2 = plan.txt_mlp
        _linear(_ln_modulate_for(w0, x_txt, t_sh2, t_sc2, xm_txt, mbs), w0, b0, ws.h[:, :Lt], gelu_from=0)
        _linear(ws.h[:, :Lt], w2, b2, x_txt, res=x_txt, gate=t_g2, gate_batch_stride=mbs)


def run_single_block(plan: _SinglePlan, ws: _Workspace, mod: Tensor, col: int, rope: _RopeTable, H: int, hd: int,
                     R: int, sp=None, x_in=None, x_out=None):
    """SingleStreamBlockProcessor.__call__ (layers.py:309-334).  linear1's output row is [q|k|v|mlp]; attention
    writes into the v slot so linear2 reads the contiguous [attn | gelu(mlp)] columns: no torch.cat.
    sp: see run_double_block — the K/V all-gather overlaps the Q and MLP-up projections.
    x_in / x_out (stand-alone processor calls): read the stream from x_in, write the block's result into x_out."""
    D = H * hd
    x_src = ws.x if x_in is None else x_in
    x_dst = ws.x if x_out is None else x_out
    y = ws.y_single(D, R)
    (shift, scale, gate), mbs = _mod_views(mod, col, 3, D)
    csb = rope.cos.stride(0) if rope.cos.shape[0] > 1 else 0
    act = _ln_modulate_for(plan.w1, x_src, shift, scale, ws.xm, mbs)
    q, k, v = y[:, :, :D], y[:, :, D: 2 * D], y[:, :, 2 * D: 3 * D]
    scales = (plan.q_scale, plan.k_scale, plan.q_scale, plan.k_scale)
    if sp is None:
        vt_ready = False
        if ws.vt_group and not plan.pv8 and (2 * D) % 256 == 0 and _bf16_operands(act, plan.w1):
            # linear1 WITHOUT its V columns (row layout [q | k | . | gelu(mlp)]) and V straight into ws.vt, one launch
            b1 = plan.b1
            vt_ready = _OPS.gemm_group([
                dict(a=act, w=plan.w1, bias=b1, out=y, gelu_from=3 * D, skip=(2 * D, D)),
                dict(x=act, w=plan.w1[2 * D: 3 * D], bias=None if b1 is None else b1[2 * D: 3 * D], vt=ws.vt, vt_pos=0, hd=hd)])
            if not vt_ready:
                ws.vt_group = False
        if not vt_ready:
            _linear(act, plan.w1, plan.b1, y, gelu_from=3 * D)
        _OPS.qknorm_rope(q, k, *scales, 0, rope.cos, rope.sin, csb, H, hd, rope.mode, q_mult=q_mult(hd))
        _joint_attention(ws, q, k, v, H, hd, plan.pv8, plan.score_bound, vt_ready)
    else:
        b1 = plan.b1
        vt_ready = False
        vt_loc = sp.local_vt(ws.B, ws.L, H, hd, ws.x.device) if (ws.vt_group and not plan.pv8 and hasattr(sp, "local_vt") and
                                                                  _bf16_operands(act, plan.w1)) else None
        if vt_loc is not None:     # (see run_double_block)
            vt_ready = _OPS.gemm_group([
                dict(a=act, w=plan.w1[D: 2 * D], bias=None if b1 is None else b1[D: 2 * D], out=y[:, :, D:]),
                dict(x=act, w=plan.w1[2 * D: 3 * D], bias=None if b1 is None else b1[2 * D: 3 * D], vt=vt_loc, vt_pos=0, hd=hd)])
            if not vt_ready:
                ws.vt_group = False
        if not vt_ready:
            _linear(act, plan.w1[D: 3 * D], None if b1 is None else b1[D: 3 * D], y[:, :, D: 3 * D])
        _OPS.qknorm_rope(None, k, *scales, 0, rope.cos, rope.sin, csb, H, hd, rope.mode)
        pending = sp.gather_kv_start(ws, k, v, H, hd, plan.pv8, vt_ready=vt_ready)
        mlp_up = lambda: _linear(act, plan.w1[3 * D:], None if b1 is None else b1[3 * D:], y[:, :, 3 * D:], gelu_from=0)
        head_mode = sp.head_parallel(H)
        if not head_mode:    # K / V^T all-gather: the MLP-up and Q projections both overlap it
            mlp_up()
        _linear(act, plan.w1[:D], None if b1 is None else b1[:D], q)
        _OPS.qknorm_rope(q, None, *scales, 0, rope.cos, rope.sin, csb, H, hd, rope.mode, q_mult=q_mult(hd))
        if head_mode:        # head exchange: q has to travel too -- project it first, the MLP-up GEMM (the block's largest) hides its all-to-all
            pending = sp.q_exchange_start(pending, q, H, hd)
            mlp_up()
        sp.attention(ws, pending, q, v, H, hd, plan.score_bound)
    _linear(y[:, :, 2 * D:], plan.w2, plan.b2, x_dst, res=x_src, gate=gate, gate_batch_stride=mbs)


def _run_modulation(vec32: Tensor, plan, D: int) -> Tensor:
    """Modulation.forward (layers.py:186-191) for the layers of one block plan sharing vec: one GEMV launch.  The task table
    (four small device tensors) is part of the plan: built once per (plan, device), not per call -- a reference model with 57
    processors installed used to pay 228 synchronous host-to-device copies per forward for it."""
    cached = getattr(plan, "_mod_tasks", None)
    if cached is None or cached[0] != str(vec32.device) or cached[1] is not _OPS:
        cols, col, task_layers = [], 0, []
        for w, b in plan.mod_layers:
            task_layers.append((w, b, col))
            cols.append(col)
            col += w.shape[0]
        cached = (str(vec32.device), _OPS, _OPS.GemvTasks(task_layers, vec32.device), cols, col)
        plan._mod_tasks = cached
    _, _, tasks, cols, ncol = cached
    mod = torch.empty(vec32.shape[0], ncol, dtype=torch.float32, device=vec32.device)
    _OPS.gemv_tasks(vec32, tasks, mod, act_in=1)
    return mod, cols


class _ProcessorPool:
    """owner of the activation workspaces of STAND-ALONE processor calls: one per (geometry, device, stream) shared by every
    block driven that way (57 blocks called one by one used to hold 57 workspaces: ADVICE r2).  Calls on one stream are ordered,
    so blocks sharing a stream can share buffers; another stream gets its own (the key carries the stream)."""


_PROC_POOL = _ProcessorPool()


def _direct_ok(*ts) -> bool:
    """can the block kernels read these caller tensors in place (LayerNorm input, residual operand of a GEMM epilogue)?  The GEMM
    ABI carries ONE set of output strides that the residual shares (osk.h: `res` "shaped like C, same strides"), and the result goes
    into a fresh contiguous [B, L, D] tensor: only a contiguous caller tensor qualifies (a strided view -- `joint[:, Lt:]`,
    a column slice -- passed the old stride-multiple test and then failed the binding's stride assertion; it is staged through the
    workspace instead, as before round 5)."""
    return all(t.dtype == BF16 and t.is_contiguous() and t.data_ptr() % 16 == 0 for t in ts)


class HipDoubleStreamBlockProcessor:
    """Drop-in for DoubleStreamBlockProcessor (layers.py:195-253): `(block, img, txt, vec, pe) -> (img, txt)`.
    bf16 callers (the reference's inference dtype): the kernels read the caller's img / txt in place (LayerNorm input, residual of
    the attention-projection GEMM) and the residual stream continues in two fresh output tensors that are returned as they are
    -- no copy in, no clone out; other dtypes are staged through the workspace."""

    def __call__(self, attn: nn.Module, img: Tensor, txt: Tensor, vec: Tensor, pe) -> tuple[Tensor, Tensor]:
        H, hd = attn.num_heads, attn.head_dim
        D = H * hd
        plan = plan_double(attn)
        B, Li, _ = img.shape
        Lt = txt.shape[1]
        R = plan.img_mlp[0].shape[0]
        ws = _workspace(_PROC_POOL, B, Lt, Li, D, R, H, hd, img.device)
        cos, sin, mode = _pe_to_cos_sin(pe, hd)
        mod, cols = _run_modulation(vec.float().contiguous(), plan, D)
        rope = _RopeTable(cos, sin, mode)
        if Li and Lt and _direct_ok(img, txt):
            o_img, o_txt = torch.empty(B, Li, D, dtype=BF16, device=img.device), torch.empty(B, Lt, D, dtype=BF16, device=img.device)
            run_double_block(plan, ws, mod, cols[0], cols[1], rope, H, hd, x_in=(img, txt), x_out=(o_img, o_txt))
            return o_img, o_txt
        ws.x[:, Lt:].copy_(img)
        ws.x[:, :Lt].copy_(txt)
        run_double_block(plan, ws, mod, cols[0], cols[1], rope, H, hd)
        return ws.x[:, Lt:].clone().to(img.dtype), ws.x[:, :Lt].clone().to(txt.dtype)


class HipSingleStreamBlockProcessor:
    """Drop-in for SingleStreamBlockProcessor (layers.py:309-334): `(block, x, vec, pe) -> x` (bf16 callers: x is read in place,
    the result is written straight into the returned tensor)."""

    def __call__(self, attn: nn.Module, x: Tensor, vec: Tensor, pe) -> Tensor:
        H, hd = attn.num_heads, attn.head_dim
        D = H * hd
        plan = plan_single(attn)
        B, L, _ = x.shape
        R = plan.w1.shape[0] - 3 * D
        ws = _workspace(_PROC_POOL, B, 0, L, D, R, H, hd, x.device)
        cos, sin, mode = _pe_to_cos_sin(pe, hd)
        mod, cols = _run_modulation(vec.float().contiguous(), plan, D)
        rope = _RopeTable(cos, sin, mode)
        if _direct_ok(x):
            out = torch.empty(B, L, D, dtype=BF16, device=x.device)
            run_single_block(plan, ws, mod, cols[0], rope, H, hd, R, x_in=x, x_out=out)
            return out
        ws.x.copy_(x)
        run_single_block(plan, ws, mod, cols[0], rope, H, hd, R)
        return ws.x.clone().to(x.dtype)


# =============================================================================================
# the model
# =============================================================================================
class MMDiTModel(_OskState, nn.Module):
    """Reference MMDiTModel (model.py:69-233) with the arithmetic in gfx950 kernels."""

    config_class = MMDiTConfig

    def __init__(self, config: MMDiTConfig):
        super().__init__()
        self.config = config
        self.in_channels = config.in_channels
        self.out_channels = self.in_channels
        self.patch_size = config.patch_size
        if config.hidden_size % config.num_heads != 0:
            raise ValueError(f"Hidden size {config.hidden_size} must be divisible by num_heads {config.num_heads}")
        pe_dim = config.hidden_size // config.num_heads
        if sum(config.axes_dim) != pe_dim:
            raise ValueError(f"Got {config.axes_dim} but expected positional dim {pe_dim}")
        if config.hidden_size % 64 != 0 or pe_dim not in (64, 72, 128):
            raise ValueError(
                f"gfx950 kernels need hidden_size % 64 == 0 and head_dim in (64, 72, 128); got {config.hidden_size}, {pe_dim}"
            )
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_heads
        self.pe_embedder = HipEmbedND(pe_dim, config.theta, config.axes_dim, liger=config.use_liger_rope)
        D = self.hidden_size
        self.img_in = nn.Linear(self.in_channels, D, bias=True)
        self.time_in = MLPEmbedder(256, D)
        self.vector_in = MLPEmbedder(config.vec_in_dim, D)
        self.guidance_in = MLPEmbedder(256, D) if config.guidance_embed else nn.Identity()
        self.cond_in = nn.Linear(self.in_channels + self.patch_size ** 2, D, bias=True) if config.cond_embed else nn.Identity()
        self.txt_in = nn.Linear(config.context_in_dim, D)
        self.double_blocks = nn.ModuleList([
            DoubleStreamBlock(D, self.num_heads, mlp_ratio=config.mlp_ratio, qkv_bias=config.qkv_bias, fused_qkv=config.fused_qkv)
            for _ in range(config.depth)])
        self.single_blocks = nn.ModuleList([
            SingleStreamBlock(D, self.num_heads, mlp_ratio=config.mlp_ratio, fused_qkv=config.fused_qkv)
            for _ in range(config.depth_single_blocks)])
        self.final_layer = LastLayer(D, 1, self.out_channels)
        if config.cond_embed:  # model.py:149-152
            nn.init.zeros_(self.cond_in.weight)
            nn.init.zeros_(self.cond_in.bias)
        self._plan = None
        self._sp = None  # set by open_sora_amd.seqpar.enable
        self.forward = self.forward_ckpt  # instance attribute, as the reference does (model.py:143-146)

    # ------------------------------------------------------------------ fp8 mode
    def enable_fp8(self, on: bool = True):
        """Opt-in reduced-precision mode (BASELINE configs[4], "fp8 MFMA"): the Linear layers of the double / single
        blocks (QKV, proj, MLP, linear1, linear2 -- >99 % of the GEMM FLOPs) run on the fp8 MFMA with per-row dynamic
        activation scales and per-output-row weight scales; attention, norms, modulation, embedders and the final
        layer stay bf16.  The reference computes in bf16: results differ by the e4m3 quantisation error."""
        for b in list(self.double_blocks) + list(self.single_blocks):
            object.__setattr__(b, "_osk_fp8", bool(on))
        self.invalidate_plan()
        return self

    # ------------------------------------------------------------------ planning
    def attention_report(self, n_seg: int = 1, seg_len: int = 0) -> dict:
        """Which attention loop body the blocks' plans select, and from which score bounds (the choice is data-dependent:
        _score_bound() reads the QK-norm scale vectors).  Blocks without a plan yet (no forward so far) are planned here."""
        hd = self.hidden_size // self.num_heads
        plans = [plan_double(b) for b in self.double_blocks] + [plan_single(b) for b in self.single_blocks]
        bounds = [float(p.score_bound) for p in plans]
        bodies = sorted({_OPS.attention_body(hd, n_seg, seg_len, b) for b in bounds}) if hasattr(_OPS, "attention_body") else []
        auto = sum(1 for b in bounds if not (0.0 < b <= SCORE_BOUND_LIMIT)) if AUTO_BOUND and hasattr(_OPS, "attention_fwd_auto") else 0
        return {"score_bound_min": min(bounds), "score_bound_max": max(bounds), "bound_limit": SCORE_BOUND_LIMIT, "bodies": bodies,
                "blocks_on_fast_body": sum(1 for b in bounds if 0.0 < b <= SCORE_BOUND_LIMIT), "blocks": len(bounds),
                # blocks whose weight-derived bound is too large: FAST / general chosen per (batch, head) on the device from the operands
                "blocks_auto_dispatched": auto}

    def invalidate_plan(self):
        self._plan = None
        for b in list(getattr(self, "double_blocks", ())) + list(getattr(self, "single_blocks", ())):
            if hasattr(b, "_osk_plan"):
                object.__delattr__(b, "_osk_plan")

    # The plan holds copies (f32 biases, concatenated / K-padded / fp8-quantised weights) and raw device pointers
    # (GemvTasks) derived from the parameters: anything that replaces or rewrites parameter storage must drop it.
    def _apply(self, fn, *a, **k):
        self.invalidate_plan()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate_plan()
        return r

    def _plan_key(self):
        """_param_key of the whole model: the plan is rebuilt when a weight's storage was swapped or the weight was written
        in place through the parameter itself (`p.copy_()` / `p.add_()` under no_grad, an optimizer step) since it was
        built.  Writes through `p.data` (a LoRA merge done as `p.data += ...`) are invisible to the version counter:
        call `invalidate_plan()` after them (see _param_key)."""
        return _param_key(self)

    def _build_plan(self, device):
        cfg = self.config
        D = self.hidden_size
        p = {}
        p["double"] = [plan_double(b) for b in self.double_blocks]
        p["single"] = [plan_single(b) for b in self.single_blocks]
        # img_in (+ cond_in) as ONE GEMM over the concatenated, K-padded input
        ws_in = [_w(self.img_in.weight)]
        b_in = self.img_in.bias.detach().float()
        if cfg.cond_embed:
            ws_in.append(_w(self.cond_in.weight))
            b_in = b_in + self.cond_in.bias.detach().float()
        p["in_w"] = _pad_k(torch.cat(ws_in, 1).contiguous())
        p["in_b"] = b_in.contiguous()
        p["txt_w"] = _pad_k(_w(self.txt_in.weight))
        p["txt_b"] = _b32(self.txt_in.bias)
        p["final_w"] = _w(self.final_layer.linear.weight)
        p["final_b"] = _b32(self.final_layer.linear.bias)
        # all adaLN layers share vec: one task list, one launch per step
        layers, col = [], 0
        p["col_double"], p["col_single"] = [], []
        for b in p["double"]:
            c = []
            for w, bias in b.mod_layers:
                layers.append((w, bias, col))
                c.append(col)
                col += w.shape[0]
            p["col_double"].append(c)
        for b in p["single"]:
            w, bias = b.mod_layers[0]
            layers.append((w, bias, col))
            p["col_single"].append(col)
            col += w.shape[0]
        fl = self.final_layer.adaLN_modulation[1]
        layers.append((_w(fl.weight), _w(fl.bias), col))
        p["col_final"] = col
        col += fl.weight.shape[0]
        p["mod_cols"] = col
        p["mod_tasks"] = _OPS.GemvTasks(layers, device)

        def emb(m):
            return (_OPS.GemvTasks([(_w(m.in_layer.weight), _w(m.in_layer.bias), 0)], device),
                    _OPS.GemvTasks([(_w(m.out_layer.weight), _w(m.out_layer.bias), 0)], device))

        p["time_in"] = emb(self.time_in)
        p["vector_in"] = emb(self.vector_in)
        p["guidance_in"] = emb(self.guidance_in) if cfg.guidance_embed else None
        p["key"] = self._plan_key()
        _watch_state_loads(self)   # a load_state_dict on any sub-module drops the plan (inference tensors: no version to key on)
        self._plan = p
        return p

    # ------------------------------------------------------------------ forward
    def prepare_block_inputs(self, img, img_ids, txt, txt_ids, timesteps, y_vec, cond=None, guidance=None,
                             _shard=None):
        """model.py:154-202.  Returns (ws, vec f32 [B, D], rope tables); img/txt land in ws.x.
        _shard = (lo, hi): embed only tokens [lo, hi) of the joint [txt;img] sequence (sequence parallelism,
        the contiguous split of distributed.py:609-624); positions stay global."""
        cfg = self.config
        if img.ndim != 3 or txt.ndim != 3:
            raise ValueError("Input img and txt tensors must have 3 dimensions.")
        if cfg.cond_embed and cond is None:
            raise ValueError("Didn't get conditional input for conditional model.")
        if cfg.guidance_embed and guidance is None:
            raise ValueError("Didn't get guidance strength for guidance distilled model.")
        dev = img.device
        p = self._plan
        if p is not None and p["key"] != self._plan_key():
            self.invalidate_plan()
            p = None
        if p is None:
            p = self._build_plan(dev)
        D, H = self.hidden_size, self.num_heads
        hd = D // H
        R = int(D * cfg.mlp_ratio)
        B = img.shape[0]
        ids = torch.cat((txt_ids, img_ids), dim=1)
        if _shard is not None:
            lo, hi = _shard
            Lt_full = txt.shape[1]
            t0, t1 = min(lo, Lt_full), min(hi, Lt_full)
            i0, i1 = max(lo, Lt_full) - Lt_full, max(hi, Lt_full) - Lt_full
            img, txt, ids = img[:, i0:i1], txt[:, t0:t1], ids[:, lo:hi]
            if cond is not None:
                cond = cond[:, i0:i1]
        Li, Lt = img.shape[1], txt.shape[1]
        ws = _workspace(self, B, Lt, Li, D, R, H, hd, dev)
        # --- img_in (+cond_in): concatenated K-padded A operand
        if Li:
            Kp = p["in_w"].shape[1]
            a_in = getattr(ws, "a_in", None)
            if a_in is None or a_in.shape[2] != Kp:
                a_in = ws.a_in = torch.zeros(B, Li, Kp, dtype=BF16, device=dev)
            C_in = img.shape[2]

            def put(src, col):   # src -> columns [col, col + C) of the K-padded operand: one osk_copy_rows_bf16 launch, no torch kernel
                if col % 4 == 0 and _OPS.copy_rows_ok(src, a_in[:, :, col:]):
                    _OPS.copy_rows(src, a_in[:, :, col:])
                else:                # (layouts the kernel does not take -- odd strides / offsets, other dtypes: torch's copy handles any)
                    a_in[:, :, col: col + src.shape[2]].copy_(src)

            put(img, 0)
            if cfg.cond_embed:
                put(cond, C_in)
            _OPS.gemm(a_in, p["in_w"], p["in_b"], ws.x[:, Lt:])
        # --- txt_in
        if Lt:
            Kt = p["txt_w"].shape[1]
            if Kt == txt.shape[2] and txt.dtype == BF16 and txt.stride(2) == 1:
                a_txt = txt
            else:
                a_txt = getattr(ws, "a_txt", None)
                if a_txt is None or a_txt.shape[2] != Kt:
                    a_txt = ws.a_txt = torch.zeros(B, Lt, Kt, dtype=BF16, device=dev)
                a_txt[:, :, : txt.shape[2]].copy_(txt)
            _OPS.gemm(a_txt, p["txt_w"], p["txt_b"], ws.x[:, :Lt])
        # --- vec = time_in(temb(t)) [+ guidance_in(temb(g))] + vector_in(y)   (f32 throughout)
        temb = torch.empty(B, 256, dtype=torch.float32, device=dev)
        hbuf = torch.empty(B, D, dtype=torch.float32, device=dev)
        vec = torch.empty(B, D, dtype=torch.float32, device=dev)
        # `t = time_factor * t` happens in the caller's dtype in the reference (layers.py:78): with the sampler's
        # bf16 t_vec, 1000*t is rounded to bf16 BEFORE the f32 sinusoid.  Keep that rounding point (B scalars).
        _OPS.timestep_embedding((1000.0 * timesteps).float().contiguous(), temb, time_factor=1.0)
        _OPS.gemv_tasks(temb, p["time_in"][0], hbuf)
        _OPS.gemv_tasks(hbuf, p["time_in"][1], vec, act_in=1)
        if cfg.guidance_embed:
            _OPS.timestep_embedding((1000.0 * guidance).float().contiguous(), temb, time_factor=1.0)
            _OPS.gemv_tasks(temb, p["guidance_in"][0], hbuf)
            _OPS.gemv_tasks(hbuf, p["guidance_in"][1], vec, act_in=1, accumulate=True)
        _OPS.gemv_tasks(y_vec.float().contiguous(), p["vector_in"][0], hbuf)
        _OPS.gemv_tasks(hbuf, p["vector_in"][1], vec, act_in=1, accumulate=True)
        # --- RoPE tables for the (local part of the) joint sequence
        rope = self.pe_embedder(ids)
        return ws, vec, rope

    def forward_ckpt(self, img: Tensor, img_ids: Tensor, txt: Tensor, txt_ids: Tensor, timesteps: Tensor,
                     y_vec: Tensor, cond: Tensor = None, guidance: Tensor | None = None, **kwargs) -> Tensor:
        """MMDiTModel.forward_ckpt (model.py:208-233); inference only.  With sequence parallelism enabled
        (open_sora_amd.seqpar.enable) this is mmdit_model_forward (distributed.py:580-683): every rank gets the
        full inputs, works on its contiguous token chunk and returns the full [B, L_img, C] prediction."""
        sp = getattr(self, "_sp", None)
        L_img, L_txt = img.shape[1], txt.shape[1]
        shard = None
        if sp is not None:
            shard = sp.shard_range(L_txt + L_img, L_txt)  # None when a rank would hold no image tokens
            if shard is None:
                sp = None
        ws, vec, rope = self.prepare_block_inputs(img, img_ids, txt, txt_ids, timesteps, y_vec, cond, guidance,
                                                  _shard=shard)
        p = self._plan
        D, H = self.hidden_size, self.num_heads
        hd = D // H
        R = int(D * self.config.mlp_ratio)
        B = img.shape[0]
        mod = torch.empty(B, p["mod_cols"], dtype=torch.float32, device=img.device)
        _OPS.gemv_tasks(vec, p["mod_tasks"], mod, act_in=1)
        for plan, (ci, ct) in zip(p["double"], p["col_double"]):
            run_double_block(plan, ws, mod, ci, ct, rope, H, hd, sp)
        for plan, c in zip(p["single"], p["col_single"]):
            run_single_block(plan, ws, mod, c, rope, H, hd, R, sp)
        # LastLayer (layers.py:398-402): (shift, scale) order
        Lt = ws.L_txt
        cf = p["col_final"]
        shift, scale = mod[:, cf: cf + D], mod[:, cf + D: cf + 2 * D]
        C_out = p["final_w"].shape[0]
        if sp is None:
            _OPS.ln_modulate(ws.x[:, Lt:], shift, scale, ws.xm[:, Lt:], mod.stride(0))
            out = torch.empty(B, ws.L_img, C_out, dtype=BF16, device=img.device)
            _OPS.gemm(ws.xm[:, Lt:], p["final_w"], p["final_b"], out)
        else:
            # equal-size gather: every rank projects ALL its rows (rank 0's few txt rows are discarded after)
            _OPS.ln_modulate(ws.x, shift, scale, ws.xm, mod.stride(0))
            out = sp.gather_output(ws, lambda dst: _OPS.gemm(ws.xm, p["final_w"], p["final_b"], dst), C_out, L_txt)
        return out.to(img.dtype) if img.dtype != BF16 else out


def Flux(cache_dir: str = None, from_pretrained: str = None, device_map="cuda", torch_dtype: torch.dtype = BF16,
         strict_load: bool = False, **kwargs) -> MMDiTModel:
    """Factory with the reference signature (model.py:271-303); `from_pretrained` takes a local safetensors / .pt path
    (open_sora_amd/ckpt.py::load_checkpoint)."""
    config = MMDiTConfig(from_pretrained=from_pretrained, cache_dir=cache_dir, **kwargs)
    with torch.device(device_map):
        model = MMDiTModel(config)
    model = model.to(torch_dtype)
    if from_pretrained:  # model.py:296-302
        from .ckpt import load_checkpoint

        model = load_checkpoint(model, from_pretrained, cache_dir=cache_dir, device_map=device_map, strict=strict_load)
    return model
