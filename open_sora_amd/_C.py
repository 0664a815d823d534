"""ctypes binding of libosk_hip.so (include/osk.h).  No torch types cross the boundary: tensors are
unwrapped to (data_ptr, strides) here; the stream is torch's current HIP stream.

There is NO fallback: if the library is missing or an entry point is absent the import fails loudly,
and a non-zero status from a kernel launch raises RuntimeError (binding layer translates C status ->
Python exception, SURVEY.md §8(b) "Error conventions").
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from .build import LIB_PATH

_i64, _i32, _f32, _f64, _vp = C.c_int64, C.c_int32, C.c_float, C.c_double, C.c_void_p

# symbol -> argtypes, exactly the declarations of include/osk.h
SIGNATURES = {
    "osk_abi_version": [],
    "osk_arch": [],
    "osk_gemm_geglu_bf16": [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp],
    "osk_attention_short_bf16": [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i32, _i32, _i32, _i32, _i32,
                                 _f32, _vp],
    "osk_attention_kernel_name": [_i32, _i32],
    "osk_attention_body_name": [_i32, _i32, _i32, _f32],
    "osk_ln_modulate_bf16": [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _vp],
    "osk_gemm_bf16": [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _i64,
                      _i32, _i32, _i32, _i32, _i32, _vp],
    "osk_gemm_bf16_pair": [_vp, _vp, _i32, _i32, _i32, _vp],     # two OskGemmOperands structs by pointer
    "osk_gemm_group_bf16": [_vp, _i32, _i32, _vp],              # OskGemmTask array by pointer
    "osk_gemm_tile_choice": [_i32, _i32, _i32],
    "osk_gemm_tile_override": [_i32],
    "osk_ln_modulate_fp8": [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _vp],
    "osk_quantize_rows_fp8": [_vp, _i64, _i64, _i32, _vp, _vp, _i32, _i32, _vp],
    "osk_gemm_fp8": [_vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _i64,
                     _i32, _i32, _i32, _i32, _i32, _vp],
    "osk_gemv_tasks_bf16": [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _vp],
    "osk_timestep_embedding": [_vp, _i32, _i32, _f32, _f32, _vp, _vp],
    "osk_rope_table": [_vp, _i64, _i32, C.POINTER(_i32), _f64, _i32, _vp, _vp, _vp],
    "osk_qknorm_rope_bf16": [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i64,
                             _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp],
    "osk_v_transpose_bf16": [_vp, _i64, _i64, _vp, _i32, _i32, _i32, _i32, _vp],
    "osk_attention_fwd_bf16": [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _vp,
                               _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _vp],
    "osk_attention_fwd_ws_bf16": [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _vp,
                                  _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _vp, _i64, _vp],
    "osk_attention_fwd_bounded_bf16": [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _vp,
                                       _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _f32, _vp, _i64, _vp],
    "osk_attention_fwd_auto_bf16": [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _vp,
                                    _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _vp, _vp, _vp, _i64, _vp],
    "osk_rownorm2_max_bf16": [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _i32, _vp],
    "osk_attention_workspace_bytes": [],
    "osk_v_scale_fp8": [_vp, _i64, _i64, _vp, _i32, _i32, _i32, _i32, _vp],
    "osk_v_transpose_fp8": [_vp, _i64, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "osk_attention_fwd_pv8_bf16": [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _vp,
                                   _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _vp, _i64, _vp],
    "osk_attention_tail_split_factor": [_i32, _i32, _i32, _i32, _i32, _i32, _i64],
    "osk_attention_launch_shape": [_i32, _i32, _i32, _i32, _i32, _i32, _f32, _i64, C.POINTER(_i32)],
    "osk_attention_rows_override": [_i32],
    "osk_cfg_euler_bf16": [_vp, _i64, _vp, _vp, _f32, _f32, _vp, _f32, _vp],
    "osk_copy_rows_bf16": [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _vp],
    "osk_causal_conv3d_ndhwc_bf16": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32,
                                     _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp],
    "osk_causal_conv3d_gn_ndhwc_bf16": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32,
                                        _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp],
    "osk_causal_conv3d_gnin_ndhwc_bf16": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _i32, _i32, _i32, _i32,
                                          _i32, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp],
    "osk_groupnorm_table_f32": [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _f32, _vp],
    "osk_groupnorm_stats_ndhwc_bf16": [_vp, _i32, _i64, _i32, _i32, _vp, _vp],
    "osk_groupnorm_apply_ndhwc_bf16": [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _f32, _i32, _vp],
    "osk_masked_softmax_f32_bf16": [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _f32, _vp],
    "osk_blend_bf16": [_vp, _vp, _i64, _i32, _i32, _i32, _i64, _vp],
    "osk_attention_hd512_fwd_ws_bf16": [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _i32, _i32,
                                        _i32, _f32, _vp, _i64, _vp],
    "osk_attention_hd512_workspace_bytes": [_i32, _i32],
    "osk_attention_hd512_fwd_bf16": [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _i32, _i32,
                                     _i32, _f32, _vp],
}


def _load() -> C.CDLL:
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m open_sora_amd.build` (hipcc, gfx950). "
            "open_sora_amd has no CPU or eager fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = (C.c_char_p if name in ("osk_arch", "osk_attention_kernel_name", "osk_attention_body_name") else
                      _i64 if name in ("osk_attention_workspace_bytes", "osk_attention_hd512_workspace_bytes") else _i32)
    if lib.osk_abi_version() != 2:
        raise ImportError("libosk_hip.so ABI version mismatch")
    return lib


class _TracingLib:
    """OSK_TRACE=1: every C-ABI call is recorded in TRACE_LOG as (entry point, arguments) before it is forwarded.  A pointer is
    replaced by 'p<k>@<low 8 bits>', k = the order in which that exact address first appeared, so two runs of the same program
    give the same log unless the SEQUENCE of launches, a scalar argument, the aliasing pattern of the buffers or an alignment
    differs (tools/diff_traces.py compares two dumps).  A debugging aid: nothing in the product path sets it."""

    def __init__(self, inner):
        self._inner = inner
        self._ids: dict = {}

    def _norm(self, kind, v):
        if kind is _vp:
            if v is None:
                return "NULL"
            v = int(v)
            k = self._ids.setdefault(v, len(self._ids))
            return f"p{k}@{v & 0xFF:02x}"
        if isinstance(v, C.Array):                 # small by-pointer arrays (osk_rope_table's axes): their contents
            return tuple(int(x) for x in v)
        return round(float(v), 9) if kind in (_f32, _f64) else int(v)

    def __getattr__(self, name):
        fn = getattr(self._inner, name)
        sig = SIGNATURES.get(name)
        if sig is None:
            return fn

        def traced(*args):
            TRACE_LOG.append((name,) + tuple(self._norm(k, a) for k, a in zip(sig, args)))
            return fn(*args)

        return traced


TRACE_LOG: list = []
lib = _load()
if os.environ.get("OSK_TRACE"):
    lib = _TracingLib(lib)


OSK_EUNSUPPORTED = -2   # csrc/osk_common.h


def _check(status: int, what: str) -> None:
    if status != 0:
        kind = "invalid argument / unsupported shape" if status < 0 else "HIP error"
        raise RuntimeError(f"{what} failed: status {status} ({kind})")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t) -> int | None:
    return None if t is None else t.data_ptr()


# ----------------------------------------------------------------------------------------------
# thin wrappers (shape logic only)
# ----------------------------------------------------------------------------------------------
def ln_modulate(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor, out: torch.Tensor,
                mod_batch_stride: int, eps: float = 1e-6) -> torch.Tensor:
    """x, out: bf16 [B, L, D] views with contiguous last dim; shift/scale f32 views whose row b starts at
    data_ptr + b*mod_batch_stride."""
    B, L, D = x.shape
    assert x.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and shift.dtype == torch.float32, (x.dtype, out.dtype)
    _check(lib.osk_ln_modulate_bf16(x.data_ptr(), x.stride(0), x.stride(1), out.data_ptr(), out.stride(0),
                                    out.stride(1), shift.data_ptr(), scale.data_ptr(), mod_batch_stride,
                                    B, L, D, eps, _stream()), "osk_ln_modulate_bf16")
    return out


# bench.py sets this to a list to collect (start, end, FLOPs) around every GEMM launch of a side measurement (never the timed region)
PROFILE_GEMM = None


class _gemm_prof:
    """context: two HIP events around the launches inside, appended to PROFILE_GEMM with their algorithmic FLOPs"""

    def __init__(self, flops: float):
        self.flops, self.on = flops, PROFILE_GEMM is not None

    def __enter__(self):
        if self.on:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            PROFILE_GEMM.append((self.e0, self.e1, self.flops))
        return False


def gemm(a: torch.Tensor, w: torch.Tensor, bias, out: torch.Tensor, *, res=None, gate=None,
         gate_batch_stride: int = 0, gelu_from: int | None = None) -> torch.Tensor:
    """a bf16 [B, L, K] view (last dim contiguous), w bf16 [N, K] (row stride arbitrary), bias f32 [N] | None,
    out bf16/f32 [B, L, N] view.  res (bf16 view shaped like out, same strides) and gate (f32, row b at
    data_ptr + b*gate_batch_stride) select the gate*x + residual epilogue."""
    B, L, K = a.shape
    N = w.shape[0]
    assert out.shape[0] == B and out.shape[1] == L and out.shape[2] == N
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and out.dtype in (torch.bfloat16, torch.float32), \
        (a.dtype, w.dtype, out.dtype)
    if res is not None:
        assert res.stride() == out.stride() and res.dtype == torch.bfloat16
    with _gemm_prof(2.0 * B * L * N * K):
        _check(lib.osk_gemm_bf16(a.data_ptr(), a.stride(0), a.stride(1), L, w.data_ptr(), w.stride(0), _p(bias),
                                 out.data_ptr(), out.stride(0), out.stride(1), L, _p(res), _p(gate),
                                 gate_batch_stride, B * L, N, K, N if gelu_from is None else gelu_from,
                                 1 if out.dtype == torch.float32 else 0, _stream()), "osk_gemm_bf16")
    return out


def geglu_pack(w_value: torch.Tensor, w_gate: torch.Tensor, b_value=None, b_gate=None):
    """Weights of a GEGLU up-projection in the row order osk_gemm_geglu_bf16 wants: value and gate rows interleaved in blocks of 16
    (include/osk.h).  w_* [N_out, K] -> [2 N_out, K]; biases f32 [N_out] -> [2 N_out] (or None).  Done once, at plan time."""
    n, k = w_value.shape
    assert w_gate.shape == (n, k) and n % 16 == 0
    w = torch.stack((w_value.reshape(n // 16, 16, k), w_gate.reshape(n // 16, 16, k)), 1).reshape(2 * n, k).contiguous()
    b = None
    if b_value is not None:
        b = torch.stack((b_value.float().reshape(n // 16, 16), b_gate.float().reshape(n // 16, 16)), 1).reshape(2 * n).contiguous()
    return w, b


def gemm_geglu(a: torch.Tensor, w_packed: torch.Tensor, bias_packed, out: torch.Tensor, workspace: torch.Tensor | None = None):
    """out[b, l, j] = value * gelu_tanh(gate) of the packed projection (geglu_pack); a bf16 [B, L, K], out bf16 [B, L, N_out].
    workspace (uint8 / bf16, >= B L 2 N_out 2 bytes): only shapes off the 256 x 256 tile path need it."""
    B, L, K = a.shape
    n_out = w_packed.shape[0] // 2
    assert out.shape == (B, L, n_out) and a.dtype == out.dtype == w_packed.dtype == torch.bfloat16
    _check(lib.osk_gemm_geglu_bf16(a.data_ptr(), a.stride(0), a.stride(1), L, w_packed.data_ptr(), w_packed.stride(0), _p(bias_packed),
                                   out.data_ptr(), out.stride(0), out.stride(1), L, B * L, n_out, K, _p(workspace),
                                   0 if workspace is None else workspace.numel() * workspace.element_size(), _stream()),
           "osk_gemm_geglu_bf16")
    return out


class OskGemmOperands(C.Structure):
    """include/osk.h::OskGemmOperands"""
    _fields_ = [("A", _vp), ("a_batch_stride", _i64), ("a_row_stride", _i64), ("a_rows_per_batch", _i32),
                ("W", _vp), ("w_row_stride", _i64), ("bias", _vp),
                ("C", _vp), ("c_batch_stride", _i64), ("c_row_stride", _i64), ("c_rows_per_batch", _i32),
                ("res", _vp), ("gate", _vp), ("gate_batch_stride", _i64), ("M", _i32)]


def _gemm_operands(a, w, bias, out, res, gate, gate_batch_stride) -> OskGemmOperands:
    B, L, K = a.shape
    assert out.shape[0] == B and out.shape[1] == L and out.shape[2] == w.shape[0] and w.shape[1] == K
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and out.dtype == torch.bfloat16
    if res is not None:
        assert res.stride() == out.stride() and res.dtype == torch.bfloat16
    return OskGemmOperands(a.data_ptr(), a.stride(0), a.stride(1), L, w.data_ptr(), w.stride(0), _p(bias), out.data_ptr(),
                           out.stride(0), out.stride(1), L, _p(res), _p(gate), gate_batch_stride, B * L)


def gemm_pair(first: dict, second: dict, *, gelu_from: int | None = None) -> None:
    """two gemm() calls that share N, K and the epilogue kind in one launch (osk_gemm_bf16_pair): each dict holds gemm()'s
    arguments a, w, bias, out and optionally res, gate, gate_batch_stride.  The img- / txt-stream Linear pairs of a double block."""
    ops = [_gemm_operands(d["a"], d["w"], d["bias"], d["out"], d.get("res"), d.get("gate"), d.get("gate_batch_stride", 0))
           for d in (first, second)]
    N, K = first["w"].shape
    assert tuple(second["w"].shape) == (N, K)
    with _gemm_prof(2.0 * (ops[0].M + ops[1].M) * N * K):
        _check(lib.osk_gemm_bf16_pair(C.addressof(ops[0]), C.addressof(ops[1]), N, K, N if gelu_from is None else gelu_from,
                                      _stream()), "osk_gemm_bf16_pair")


class OskGemmTask(C.Structure):
    """include/osk.h::OskGemmTask"""
    _fields_ = [("op", OskGemmOperands), ("N", _i32), ("gelu_from", _i32), ("skip_from", _i32), ("skip_len", _i32), ("vt_head_dim", _i32)]


def gemm_group(tasks: list) -> bool:
    """osk_gemm_group_bf16: up to four problems sharing K in one launch.  Each task is a dict:
      plain:  a, w, bias, out [, res, gate, gate_batch_stride, gelu_from, skip=(from, len)]  -- gemm()'s arguments; with `skip` the
              physical columns [from, from + len) of w / bias / out are neither computed nor stored
      V^T:    x (bf16 [B, L, K] view), w (W_v [H*hd, K]), bias (f32 | None), vt (bf16 [B, H, hd, Lp] contiguous), vt_pos (first position on
              the key axis, % 64 == 0), hd
    Returns False -- nothing launched -- when the library declines the group (OSK_EUNSUPPORTED: a task off the 256 x 256 tile path);
    the caller then runs the single calls."""
    arr = (OskGemmTask * len(tasks))()
    K = None
    flops = 0.0
    for t, d in zip(arr, tasks):
        if "vt" in d:
            x, w, vt = d["x"], d["w"], d["vt"]
            B, L, k_ = x.shape
            assert x.dtype == w.dtype == vt.dtype == torch.bfloat16 and vt.is_contiguous() and vt.dim() == 4 and d["vt_pos"] % 64 == 0
            assert vt.shape[0] == B and vt.shape[1] * vt.shape[2] == w.shape[0] and d["vt_pos"] + (L + 63) // 64 * 64 <= vt.shape[3]
            t.op = OskGemmOperands(x.data_ptr(), x.stride(0), x.stride(1), L, w.data_ptr(), w.stride(0), _p(d.get("bias")),
                                   vt.data_ptr() + 2 * d["vt_pos"], vt.stride(0), vt.stride(2), L, None, None, 0, B * L)
            t.N, t.gelu_from, t.skip_from, t.skip_len, t.vt_head_dim = w.shape[0], w.shape[0], 0, 0, d["hd"]
            flops += 2.0 * B * L * w.shape[0] * k_
        else:
            a, w, out = d["a"], d["w"], d["out"]
            k_ = a.shape[2]
            t.op = _gemm_operands_n(a, w, d.get("bias"), out, d.get("res"), d.get("gate"), d.get("gate_batch_stride", 0))
            sf, sl = d.get("skip", (0, 0))
            t.N, t.gelu_from, t.skip_from, t.skip_len, t.vt_head_dim = w.shape[0], d.get("gelu_from", w.shape[0]) if d.get("gelu_from") is not None else w.shape[0], sf, sl, 0
            flops += 2.0 * a.shape[0] * a.shape[1] * (w.shape[0] - sl) * k_
        assert K in (None, k_), "the tasks of a group share K"
        K = k_
    with _gemm_prof(flops):
        rc = lib.osk_gemm_group_bf16(C.addressof(arr), len(tasks), K, _stream())
    if rc == OSK_EUNSUPPORTED:
        return False
    _check(rc, "osk_gemm_group_bf16")
    return True


def _gemm_operands_n(a, w, bias, out, res, gate, gate_batch_stride) -> OskGemmOperands:
    """_gemm_operands for a task whose output tensor may be WIDER than the task's N (a skip range / a column slice: the row layout
    [q | k | . | mlp] is addressed through its row stride)"""
    B, L, K = a.shape
    assert out.shape[0] == B and out.shape[1] == L and out.shape[2] >= 1 and w.shape[1] == K
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and out.dtype == torch.bfloat16
    if res is not None:
        assert res.stride() == out.stride() and res.dtype == torch.bfloat16
    return OskGemmOperands(a.data_ptr(), a.stride(0), a.stride(1), L, w.data_ptr(), w.stride(0), _p(bias), out.data_ptr(),
                           out.stride(0), out.stride(1), L, _p(res), _p(gate), gate_batch_stride, B * L)


def ln_modulate_fp8(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor, mod_batch_stride: int, eps: float = 1e-6):
    """ln_modulate + quantize_rows_fp8 in one pass: x bf16 [B, L, D] view -> (e4m3 bytes uint8 [B*L, D], f32 scales [B*L])."""
    B, L, D = x.shape
    out8 = torch.empty(B * L, D, dtype=torch.uint8, device=x.device)
    scales = torch.empty(B * L, dtype=torch.float32, device=x.device)
    _check(lib.osk_ln_modulate_fp8(x.data_ptr(), x.stride(0), x.stride(1), out8.data_ptr(), scales.data_ptr(),
                                   shift.data_ptr(), scale.data_ptr(), mod_batch_stride, B, L, D, eps, _stream()),
           "osk_ln_modulate_fp8")
    return out8, scales


def quantize_rows_fp8(x: torch.Tensor, out8: torch.Tensor | None = None, scales: torch.Tensor | None = None):
    """x bf16 [B, L, K] view (or [N, K] weight) -> (e4m3 bytes as uint8 [B*L, K] contiguous, f32 scales [B*L])."""
    if x.dim() == 2:
        x = x.unsqueeze(0)
    B, L, K = x.shape
    M = B * L
    if out8 is None:
        out8 = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    if scales is None:
        scales = torch.empty(M, dtype=torch.float32, device=x.device)
    assert out8.is_contiguous() and out8.numel() == M * K and scales.numel() >= M
    _check(lib.osk_quantize_rows_fp8(x.data_ptr(), x.stride(0), x.stride(1), L, out8.data_ptr(), scales.data_ptr(),
                                     M, K, _stream()), "osk_quantize_rows_fp8")
    return out8, scales


def gemm_fp8_supported(M: int, N: int, K: int) -> bool:
    """shapes the fp8 instantiation of the large-tile kernel takes (include/osk.h); others stay on gemm().
    Mirrors gemm256_fp8_supported (csrc/gemm256.hip): the kernel's 32-bit per-lane source offsets need both the
    (contiguous, as the host passes them) activation and weight images to span < 4 GiB."""
    return M >= 256 and N >= 128 and K % 128 == 0 and M * K < 0xFFFFFFFF and N * K < 0xFFFFFFFF


def gemm_fp8(a8: torch.Tensor, a_scale: torch.Tensor, w8: torch.Tensor, w_scale: torch.Tensor, bias,
             out: torch.Tensor, *, res=None, gate=None, gate_batch_stride: int = 0,
             gelu_from: int | None = None) -> torch.Tensor:
    """a8 uint8 (e4m3) [M, K] contiguous with scales f32 [M]; w8 uint8 [N, K] with scales f32 [N]; out bf16/f32
    [B, L, N] view with B * L == M; epilogue arguments as gemm()."""
    M, K = a8.shape
    N = w8.shape[0]
    B, L = out.shape[0], out.shape[1]
    assert B * L == M and out.shape[2] == N and w8.shape[1] == K
    if res is not None:
        assert res.stride() == out.stride()
    with _gemm_prof(2.0 * M * N * K):
        _check(lib.osk_gemm_fp8(a8.data_ptr(), 0, a8.stride(0), M, a_scale.data_ptr(), w8.data_ptr(), w8.stride(0),
                                w_scale.data_ptr(), _p(bias), out.data_ptr(), out.stride(0), out.stride(1), L, _p(res),
                                _p(gate), gate_batch_stride, M, N, K, N if gelu_from is None else gelu_from,
                                1 if out.dtype == torch.float32 else 0, _stream()), "osk_gemm_fp8")
    return out


def gemv_tasks(x: torch.Tensor, tasks, out: torch.Tensor, act_in: int = 0, accumulate: bool = False):
    """tasks: GemvTasks (device descriptor arrays).  x f32 [Bv, K], out f32 [Bv, *]."""
    Bv, K = x.shape
    _check(lib.osk_gemv_tasks_bf16(x.data_ptr(), x.stride(0), Bv, K, tasks.w_ptrs.data_ptr(),
                                   tasks.b_ptrs.data_ptr(), tasks.out_cols.data_ptr(),
                                   tasks.n_rows.data_ptr(), tasks.n_tasks, out.data_ptr(), out.stride(0),
                                   act_in, 1 if accumulate else 0, _stream()), "osk_gemv_tasks_bf16")
    return out


class GemvTasks:
    """Device-side task list for osk_gemv_tasks_bf16: a list of (weight bf16 [N, K], bias bf16 [N] | None,
    out column offset); each layer is cut into tasks of <= 64 rows.  Keeps the tensors alive."""

    ROWS = 64

    def __init__(self, layers, device):
        wp, bp, oc, nr = [], [], [], []
        self._keep = []
        for w, b, col in layers:
            assert w.dtype == torch.bfloat16 and w.is_contiguous()
            self._keep.append((w, b))
            N, K = w.shape
            for r0 in range(0, N, self.ROWS):
                n = min(self.ROWS, N - r0)
                wp.append(w.data_ptr() + r0 * K * 2)
                bp.append(0 if b is None else b.data_ptr() + r0 * 2)
                oc.append(col + r0)
                nr.append(n)
        self.n_tasks = len(wp)
        # uint64 pointers stored bit-exactly in int64 tensors
        to_i64 = lambda v: v - (1 << 64) if v >= (1 << 63) else v
        self.w_ptrs = torch.tensor([to_i64(v) for v in wp], dtype=torch.int64, device=device)
        self.b_ptrs = torch.tensor([to_i64(v) for v in bp], dtype=torch.int64, device=device)
        self.out_cols = torch.tensor(oc, dtype=torch.int32, device=device)
        self.n_rows = torch.tensor(nr, dtype=torch.int32, device=device)


def timestep_embedding(t: torch.Tensor, out: torch.Tensor, max_period: float = 10000.0,
                       time_factor: float = 1000.0) -> torch.Tensor:
    B, dim = out.shape
    _check(lib.osk_timestep_embedding(t.data_ptr(), B, dim, max_period, time_factor, out.data_ptr(), _stream()),
           "osk_timestep_embedding")
    return out


def rope_table(ids: torch.Tensor, axes_dim, theta: float, f32_angles: bool, cos: torch.Tensor, sin: torch.Tensor):
    """ids f32 [n_rows, n_axes] contiguous -> cos/sin f32 [n_rows, sum(axes)/2]."""
    n_rows, n_axes = ids.shape
    arr = (_i32 * n_axes)(*[int(a) for a in axes_dim])
    _check(lib.osk_rope_table(ids.data_ptr(), n_rows, n_axes, arr, float(theta), 1 if f32_angles else 0,
                              cos.data_ptr(), sin.data_ptr(), _stream()), "osk_rope_table")


def qknorm_rope(q: torch.Tensor, k: torch.Tensor, qs0, ks0, qs1, ks1, l_split: int, cos, sin,
                cs_batch_stride: int, H: int, hd: int, rope_mode: int, eps: float = 1e-6, q_mult: float = 1.0):
    """q, k: bf16 [B, L, H*hd] views (same strides, last dim contiguous) rewritten in place; one of them may
    be None (one-sided call).  q_mult: factor folded into q before its final rounding (attention_fwd q_prescaled)."""
    t = q if q is not None else k
    B, L, _ = t.shape
    assert q is None or k is None or q.stride() == k.stride()
    _check(lib.osk_qknorm_rope_bf16(_p(q), _p(k), t.stride(0), t.stride(1), qs0.data_ptr(),
                                    ks0.data_ptr(), qs1.data_ptr(), ks1.data_ptr(), l_split, cos.data_ptr(),
                                    sin.data_ptr(), cs_batch_stride, B, L, H, hd, rope_mode, eps, q_mult, _stream()),
           "osk_qknorm_rope_bf16")


def v_transpose(v: torch.Tensor, vt: torch.Tensor, H: int, hd: int):
    """v bf16 [B, L, H*hd] view -> vt bf16 [B, H, hd, round_up(L, 64)] contiguous."""
    B, L, _ = v.shape
    _check(lib.osk_v_transpose_bf16(v.data_ptr(), v.stride(0), v.stride(1), vt.data_ptr(), B, L, H, hd, _stream()),
           "osk_v_transpose_bf16")


# bench.py sets this to a list to collect (start, end) HIP events around every attention launch
PROFILE_ATTENTION = None


_ATTN_WS: dict = {}


def attention_workspace(device) -> torch.Tensor:
    """workspace for the tail split of osk_attention_fwd_ws_bf16, one per (device, stream): launches are ordered within a
    stream (the merge kernel of one launch has consumed the buffer before the next launch writes it), not across
    streams.  Allocated once per key, reused by every launch."""
    dev = torch.device(device)
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
    ws = _ATTN_WS.get(key)
    if ws is None:
        ws = _ATTN_WS[key] = torch.empty(lib.osk_attention_workspace_bytes(), dtype=torch.uint8, device=dev)
    return ws


def attention_fwd(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, H: int, hd: int,
                  scale: float, *, lse=None, n_seg: int = 1, seg_len: int | None = None,
                  k_seg_stride: int = 0, vt_seg_stride: int = 0, q_prescaled: bool = False, kv_batches: int = 0,
                  workspace: torch.Tensor | None = None, score_bound: float = 0.0):
    """q bf16 [B, Lq, H*hd] view; k bf16 [B, seg_len, H*hd] view of segment 0 (further segments k_seg_stride
    elements apart); vt from v_transpose (per segment); out bf16 [B, Lq, H*hd] view.  workspace (uint8, from
    attention_workspace()): lets the library split the workgroups of the grid's last partial round along the keys.
    score_bound > 0: |q . k| (as the kernel sees the scores, log2 units) never exceeds it -- enables the fast body
    (include/osk.h, osk_attention_fwd_bounded_bf16)."""
    B, Lq, _ = q.shape
    if seg_len is None:
        seg_len = k.shape[1]
    if CHECK_SCORE_BOUND and score_bound > 0 and not torch.cuda.is_current_stream_capturing():   # (the check syncs: illegal in a capture)
        _assert_score_bound(q, k, H, hd, scale, n_seg, seg_len, k_seg_stride, q_prescaled, kv_batches, score_bound)
    prof = PROFILE_ATTENTION
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _check(lib.osk_attention_fwd_bounded_bf16(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k_seg_stride,
                                              k.stride(0), k.stride(1), vt.data_ptr(), vt_seg_stride, out.data_ptr(),
                                              out.stride(0), out.stride(1), _p(lse), B, H, Lq, n_seg, seg_len, hd,
                                              scale, int(q_prescaled), kv_batches, float(score_bound), _p(workspace),
                                              0 if workspace is None else workspace.numel(), _stream()),
           "osk_attention_fwd_bounded_bf16")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1))
    return out


def rownorm2_max(x: torch.Tensor, out: torch.Tensor, H: int, hd: int, accumulate: bool = False) -> torch.Tensor:
    """out[b, h] = max_l |x[b, l, h, :]|^2 of a bf16 [B, L, H*hd] view (osk_rownorm2_max_bf16); out f32 [B, H] contiguous"""
    B, L, _ = x.shape
    assert x.dtype == torch.bfloat16 and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == B * H
    _check(lib.osk_rownorm2_max_bf16(x.data_ptr(), x.stride(0), x.stride(1), B, L, H, hd, out.data_ptr(), 1 if accumulate else 0, _stream()),
           "osk_rownorm2_max_bf16")
    return out


def attention_fwd_auto(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, H: int, hd: int, scale: float,
                       qn2: torch.Tensor, kn2: torch.Tensor, *, lse=None, n_seg: int = 1, seg_len: int | None = None,
                       k_seg_stride: int = 0, vt_seg_stride: int = 0, q_prescaled: bool = True, kv_batches: int = 0,
                       workspace: torch.Tensor | None = None):
    """attention_fwd with the score bound taken from the operands on the device (osk_attention_fwd_auto_bf16): qn2 / kn2 from
    rownorm2_max() of the q / k of THIS call (f32 [B, H] / [kv_batches or B, H]).  Units whose bound allows it run the FAST body."""
    B, Lq, _ = q.shape
    if seg_len is None:
        seg_len = k.shape[1]
    assert qn2.dtype == kn2.dtype == torch.float32 and qn2.numel() == B * H and kn2.numel() == (kv_batches or B) * H
    prof = PROFILE_ATTENTION
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _check(lib.osk_attention_fwd_auto_bf16(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k_seg_stride,
                                           k.stride(0), k.stride(1), vt.data_ptr(), vt_seg_stride, out.data_ptr(),
                                           out.stride(0), out.stride(1), _p(lse), B, H, Lq, n_seg, seg_len, hd,
                                           scale, int(q_prescaled), kv_batches, qn2.data_ptr(), kn2.data_ptr(), _p(workspace),
                                           0 if workspace is None else workspace.numel(), _stream()),
           "osk_attention_fwd_auto_bf16")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1))
    return out


def attention_short(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, H: int, hd: int, scale: float,
                    alibi_slopes: torch.Tensor | None = None) -> torch.Tensor:
    """softmax(scale q k^T - slope_h |i + Lk - Lq - j|) v for sequences of at most 64 tokens (osk_attention_short_bf16: the
    temporal-attention call shape); q [B, Lq, H*hd], k / v [B, Lk, H*hd] bf16 views, alibi_slopes f32 [H] or None."""
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    if alibi_slopes is not None:
        assert alibi_slopes.dtype == torch.float32 and alibi_slopes.is_contiguous() and alibi_slopes.numel() == H
    _check(lib.osk_attention_short_bf16(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1), v.data_ptr(),
                                        v.stride(0), v.stride(1), out.data_ptr(), out.stride(0), out.stride(1), _p(alibi_slopes),
                                        B, H, Lq, Lk, hd, scale, _stream()), "osk_attention_short_bf16")
    return out


def attention_launch_shape(B: int, H: int, Lq: int, n_seg: int, seg_len: int, hd: int, score_bound: float, workspace_bytes: int):
    """(key parts of the last round's work units, query rows per work unit) of a bounded attention call with these arguments"""
    rows = _i32(0)
    parts = lib.osk_attention_launch_shape(B, H, Lq, n_seg, seg_len, hd, float(score_bound), int(workspace_bytes), C.byref(rows))
    return int(parts), int(rows.value)


def attention_body(hd: int, n_seg: int, seg_len: int, score_bound: float) -> str:
    """which loop body attention_fwd(..., score_bound=...) runs for this key layout (reporting: bench.py, tests)"""
    return lib.osk_attention_body_name(hd, n_seg, seg_len, float(score_bound)).decode()


# OSK_CHECK_SCORE_BOUND=1 (debugging runs; its own switch since round 5 -- the call tracer OSK_TRACE only records calls, as its
# docstring says): every bounded attention call checks the caller's promise on the device before it launches -- the bounded loop
# body has no running maximum, so a violated bound silently loses accuracy (or, far beyond it, overflows).  The check is the
# sufficient condition the model's own bound is derived from (Cauchy-Schwarz per head): max |q_h| * max |k_h| <= bound.  It reads a
# device scalar back (a blocking sync), so it is skipped while the stream is being captured into a hipGraph.
CHECK_SCORE_BOUND = bool(os.environ.get("OSK_CHECK_SCORE_BOUND"))


def _assert_score_bound(q, k, H, hd, scale, n_seg, seg_len, k_seg_stride, q_prescaled, kv_batches, bound):
    Bq, Lq, _ = q.shape
    Bk = kv_batches if kv_batches else Bq
    qn = q.float().reshape(Bq, Lq, H, hd).norm(dim=-1).amax()
    if not q_prescaled:
        qn = qn * (scale * 1.4426950408889634)
    kn = torch.zeros((), dtype=torch.float32, device=q.device)
    for s_ in range(n_seg):
        ks = torch.as_strided(k, (Bk, seg_len, H * hd), (k.stride(0), k.stride(1), 1), k.storage_offset() + s_ * k_seg_stride)
        kn = torch.maximum(kn, ks.float().reshape(Bk, seg_len, H, hd).norm(dim=-1).amax())
    worst = float(qn * kn)
    if worst > bound * (1 + 2.0 ** -7):
        raise RuntimeError(f"osk_attention_fwd_bounded_bf16: the caller's score bound {bound:.4g} does not hold: max |q| max |k| = {worst:.4g} "
                           "(log2 units) -- the QK-norm scale vectors changed after the plan was built?")


def vt8_rows(hd: int) -> int:
    """rows per head of the e4m3 V^T tensor: hd dims + the ones row, rounded up to 16"""
    return (hd + 1 + 15) // 16 * 16


def v_scale_fp8(v: torch.Tensor, H: int, hd: int) -> torch.Tensor:
    """v bf16 [B, L, H*hd] view -> e4m3 scales f32 [B, H] = absmax per (batch, head) / 448"""
    B, L, _ = v.shape
    scales = torch.empty(B, H, dtype=torch.float32, device=v.device)
    _check(lib.osk_v_scale_fp8(v.data_ptr(), v.stride(0), v.stride(1), scales.data_ptr(), B, L, H, hd, _stream()),
           "osk_v_scale_fp8")
    return scales


def v_transpose_fp8(v: torch.Tensor, scales: torch.Tensor, vt8: torch.Tensor, H: int, hd: int) -> torch.Tensor:
    """v bf16 [B, L, H*hd] view, scales f32 [B, H] -> vt8 uint8 [B, H, vt8_rows(hd), Lp] (e4m3 bytes, kernel key order)."""
    B, L, _ = v.shape
    assert vt8.dtype == torch.uint8 and vt8.is_contiguous() and vt8.shape[-2] == vt8_rows(hd)
    assert scales.dtype == torch.float32 and scales.is_contiguous() and scales.numel() == B * H
    _check(lib.osk_v_transpose_fp8(v.data_ptr(), v.stride(0), v.stride(1), scales.data_ptr(), vt8.data_ptr(), B, L, H, hd,
                                   _stream()), "osk_v_transpose_fp8")
    return vt8


def attention_fwd_pv8(q: torch.Tensor, k: torch.Tensor, vt8: torch.Tensor, v_scale: torch.Tensor, out: torch.Tensor,
                      H: int, hd: int, scale: float, *, lse=None, n_seg: int = 1, seg_len: int | None = None,
                      k_seg_stride: int = 0, vt_seg_stride: int = 0, q_prescaled: bool = False, kv_batches: int = 0,
                      workspace: torch.Tensor | None = None):
    """attention_fwd with the P.V product on the fp8 MFMA: vt8 / v_scale from v_transpose_fp8 (vt_seg_stride in bytes)."""
    B, Lq, _ = q.shape
    if seg_len is None:
        seg_len = k.shape[1]
    prof = PROFILE_ATTENTION
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _check(lib.osk_attention_fwd_pv8_bf16(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k_seg_stride,
                                          k.stride(0), k.stride(1), vt8.data_ptr(), vt_seg_stride, v_scale.data_ptr(),
                                          out.data_ptr(), out.stride(0), out.stride(1), _p(lse), B, H, Lq, n_seg,
                                          seg_len, hd, scale, int(q_prescaled), kv_batches, _p(workspace),
                                          0 if workspace is None else workspace.numel(), _stream()),
           "osk_attention_fwd_pv8_bf16")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1))
    return out


def copy_rows_ok(src: torch.Tensor, dst: torch.Tensor) -> bool:
    """every precondition of osk_copy_rows_bf16 (csrc/elementwise.hip): bf16, unit channel stride, C % 4 == 0, all other strides
    % 4 == 0, 8-byte aligned pointers, non-empty -- callers fall back to torch's copy otherwise"""
    if not (src.dtype == dst.dtype == torch.bfloat16 and src.ndim == dst.ndim and src.ndim in (3, 4)):
        return False
    if src.numel() == 0 or src.shape[-1] % 4 or src.stride(-1) != 1 or dst.stride(-1) != 1 or dst.shape[-1] < src.shape[-1]:
        return False
    if any(st % 4 for st in src.stride()[:-1]) or any(st % 4 for st in dst.stride()[:-1]):
        return False
    return src.data_ptr() % 8 == 0 and dst.data_ptr() % 8 == 0


def copy_rows(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst[..., :C] = src for bf16 [B, L, C] -- or [P, B, L, C], P chunks -- views with unit channel stride on both sides (dst rows
    may be wider than C: a column slice of a K-padded operand; either side's chunk axis may be a column group of a token-major
    tensor: `x.view(B, L, P, C).permute(2, 0, 1, 3)`); a 3-D src of batch 1 -- or an expanded view -- is broadcast over dst's batch."""
    assert src.dtype == dst.dtype == torch.bfloat16 and src.ndim == dst.ndim and src.ndim in (3, 4), (src.dtype, dst.dtype, src.shape, dst.shape)
    if src.ndim == 3:
        src, dst = src.unsqueeze(0), dst.unsqueeze(0)
    NC, B, L, Cc = dst.shape[0], dst.shape[1], dst.shape[2], src.shape[3]
    assert src.shape[0] == NC and src.shape[2] == L and src.shape[1] in (1, B) and dst.shape[3] >= Cc and src.stride(3) == 1 and dst.stride(3) == 1
    sbs = 0 if src.shape[1] == 1 else src.stride(1)
    _check(lib.osk_copy_rows_bf16(src.data_ptr(), src.stride(0), sbs, src.stride(2), dst.data_ptr(), dst.stride(0), dst.stride(1), dst.stride(2),
                                  NC, B, L, Cc, _stream()), "osk_copy_rows_bf16")
    return dst


def cfg_euler(pred: torch.Tensor, x: torch.Tensor, x_out: torch.Tensor, g_txt: float, g_img: float, dt: float,
              g_img_vec=None):
    """pred bf16 [3, ...] contiguous (cond, uncond, uncond_2); x, x_out bf16 [...] contiguous."""
    n = x.numel()
    assert pred.numel() == 3 * n and pred.is_contiguous() and x.is_contiguous() and x_out.is_contiguous()
    assert pred.dtype == x.dtype == x_out.dtype == torch.bfloat16, \
        f"osk_cfg_euler_bf16 reads and writes bf16; got {pred.dtype}, {x.dtype}, {x_out.dtype}"
    _check(lib.osk_cfg_euler_bf16(pred.data_ptr(), n, x.data_ptr(), x_out.data_ptr(), g_txt, g_img,
                                  _p(g_img_vec), dt, _stream()), "osk_cfg_euler_bf16")
    return x_out


# ----------------------------------------------------------------------------------------------
# causal 3-D VAE kernels (NDHWC bf16)
# ----------------------------------------------------------------------------------------------
# bench.py sets this to a list to collect (start, end, padded-channel FLOPs) around every conv launch
PROFILE_CONV = None


def conv_out_dims(T: int, H: int, W: int, stride=(1, 1, 1), up=(False, False)):
    Tu = 1 + 2 * (T - 1) if up[0] else T
    Hu, Wu = (2 * H, 2 * W) if up[1] else (H, W)
    return (Tu - 1) // stride[0] + 1, (Hu - 1) // stride[1] + 1, (Wu - 1) // stride[2] + 1


def causal_conv3d(x: torch.Tensor, w: torch.Tensor, bias, out: torch.Tensor, ksize: int, stride=(1, 1, 1),
                  up=(False, False), res=None, gn_sums: torch.Tensor | None = None) -> torch.Tensor:
    """x bf16 [B, T, H, W, Cin] contiguous; w bf16 [Cout, Kpad] (tap-major, channel-minor, zero padded);
    bias f32 [Cout] | None; out bf16 [B, To, Ho, Wo, Cout] contiguous; res like out | None.
    gn_sums (f64 [B, G, 2], zeroed by the caller): also accumulate the GroupNorm statistics of `out` in the conv's
    epilogue; the return value is then (out, fused).  The kernels that carry that epilogue do not take every shape: when
    they do not, the plain conv runs, gn_sums stays untouched and fused is False (the consumer runs groupnorm_stats)."""
    B, T, H, W, Cin = x.shape
    Cout = w.shape[0]
    To, Ho, Wo = conv_out_dims(T, H, W, stride, up)
    assert x.is_contiguous() and out.is_contiguous() and tuple(out.shape) == (B, To, Ho, Wo, Cout), (out.shape, (B, To, Ho, Wo, Cout))
    assert res is None or (res.is_contiguous() and res.shape == out.shape)
    prof = PROFILE_CONV
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    fused = False
    if gn_sums is not None:
        assert gn_sums.dtype == torch.float64 and gn_sums.is_contiguous() and gn_sums.shape[0] == B and gn_sums.shape[2] == 2
        rc = lib.osk_causal_conv3d_gn_ndhwc_bf16(x.data_ptr(), B, T, H, W, Cin, w.data_ptr(), w.stride(0), _p(bias), Cout,
                                                 ksize, stride[0], stride[1], stride[2], int(up[0]), int(up[1]), _p(res),
                                                 out.data_ptr(), To, Ho, Wo, gn_sums.data_ptr(), gn_sums.shape[1], _stream())
        fused = rc == 0
        if rc != 0 and rc != OSK_EUNSUPPORTED:
            _check(rc, "osk_causal_conv3d_gn_ndhwc_bf16")
    if not fused:
        _check(lib.osk_causal_conv3d_ndhwc_bf16(x.data_ptr(), B, T, H, W, Cin, w.data_ptr(), w.stride(0), _p(bias), Cout,
                                                ksize, stride[0], stride[1], stride[2], int(up[0]), int(up[1]), _p(res),
                                                out.data_ptr(), To, Ho, Wo, _stream()), "osk_causal_conv3d_ndhwc_bf16")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 2.0 * Cin * Cout * ksize ** 3 * B * To * Ho * Wo))
    return out if gn_sums is None else (out, fused)


def causal_conv3d_gn_in(x: torch.Tensor, table: torch.Tensor, w: torch.Tensor, bias, out: torch.Tensor, ksize: int,
                        stride=(1, 1, 1), res=None, gn_sums: torch.Tensor | None = None):
    """conv(silu(GroupNorm(x))) reading x itself: `table` (groupnorm_table) carries the norm's per-(batch, channel) scale / shift and
    the sliding-window kernels apply it -- with the rounding points of groupnorm_apply(silu=True) -- while they refill their halo.
    -> (ran, fused): ran False = the shape is not one those kernels take and NOTHING was launched (the caller runs
    groupnorm_apply + causal_conv3d); fused = gn_sums was accumulated in the epilogue as in causal_conv3d."""
    B, T, H, W, Cin = x.shape
    Cout = w.shape[0]
    To, Ho, Wo = conv_out_dims(T, H, W, stride)
    assert x.is_contiguous() and out.is_contiguous() and tuple(out.shape) == (B, To, Ho, Wo, Cout), (out.shape, (B, To, Ho, Wo, Cout))
    assert res is None or (res.is_contiguous() and res.shape == out.shape)
    assert table.dtype == torch.float32 and table.is_contiguous() and tuple(table.shape) == (B, Cin // 8, 16), table.shape
    prof = PROFILE_CONV
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()

    def call(sums):
        return lib.osk_causal_conv3d_gnin_ndhwc_bf16(x.data_ptr(), table.data_ptr(), B, T, H, W, Cin, w.data_ptr(), w.stride(0),
                                                     _p(bias), Cout, ksize, stride[0], stride[1], stride[2], _p(res),
                                                     out.data_ptr(), To, Ho, Wo, _p(sums), 0 if sums is None else sums.shape[1],
                                                     _stream())

    fused = False
    rc = OSK_EUNSUPPORTED
    if gn_sums is not None:
        assert gn_sums.dtype == torch.float64 and gn_sums.is_contiguous() and gn_sums.shape[0] == B and gn_sums.shape[2] == 2
        rc = call(gn_sums)
        fused = rc == 0
    if rc == OSK_EUNSUPPORTED:
        rc = call(None)
    if rc == OSK_EUNSUPPORTED:
        return False, False
    _check(rc, "osk_causal_conv3d_gnin_ndhwc_bf16")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 2.0 * Cin * Cout * ksize ** 3 * B * To * Ho * Wo))
    return True, fused


def groupnorm_table(sums: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, table: torch.Tensor, S: int, G: int,
                    eps: float = 1e-6) -> torch.Tensor:
    """sums f64 [B, G, 2] over S voxels -> table f32 [B, C / 8, 16]: per 8-channel chunk 8 scales rstd gamma, 8 shifts beta - mean
    rstd gamma (the constants groupnorm_apply derives), for causal_conv3d_gn_in."""
    B, C = sums.shape[0], gamma.numel()
    assert table.dtype == torch.float32 and table.is_contiguous() and tuple(table.shape) == (B, C // 8, 16), table.shape
    _check(lib.osk_groupnorm_table_f32(sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), table.data_ptr(), B, S, C, G, eps,
                                       _stream()), "osk_groupnorm_table_f32")
    return table


def groupnorm_stats(x: torch.Tensor, G: int, sums: torch.Tensor) -> torch.Tensor:
    """x bf16 [B, ..., C] contiguous -> sums f64 [B, G, 2]."""
    B, C = x.shape[0], x.shape[-1]
    S = x.numel() // (B * C)
    _check(lib.osk_groupnorm_stats_ndhwc_bf16(x.data_ptr(), B, S, C, G, sums.data_ptr(), _stream()),
           "osk_groupnorm_stats_ndhwc_bf16")
    return sums


def groupnorm_apply(x: torch.Tensor, sums: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out: torch.Tensor,
                    G: int, eps: float = 1e-6, silu: bool = True) -> torch.Tensor:
    B, C = x.shape[0], x.shape[-1]
    S = x.numel() // (B * C)
    _check(lib.osk_groupnorm_apply_ndhwc_bf16(x.data_ptr(), sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                              out.data_ptr(), B, S, C, G, eps, 1 if silu else 0, _stream()),
           "osk_groupnorm_apply_ndhwc_bf16")
    return out


def masked_softmax(scores: torch.Tensor, probs: torch.Tensor, Sk: int, keys_per_frame: int, scale: float):
    """scores f32 [Sq, >=Sk] (row stride arbitrary), probs bf16 [Sq, ldp] contiguous rows, ldp >= Sk."""
    Sq = scores.shape[0]
    _check(lib.osk_masked_softmax_f32_bf16(scores.data_ptr(), scores.stride(0), probs.data_ptr(), probs.stride(0), Sq,
                                           Sk, keys_per_frame, scale, _stream()), "osk_masked_softmax_f32_bf16")
    return probs


def blend(a: torch.Tensor, b: torch.Tensor, extent: int, dim: int) -> torch.Tensor:
    """tile cross-fade along `dim`, written into b: the last `extent` slices of a fade into the first `extent` of b.
    a, b: contiguous bf16 tensors that agree in every other dimension."""
    dim = dim % b.ndim
    extent = min(a.shape[dim], b.shape[dim], extent)
    if extent == 0:
        return b
    assert a.dtype == b.dtype == torch.bfloat16 and a.is_contiguous() and b.is_contiguous()
    assert a.shape[:dim] == b.shape[:dim] and a.shape[dim + 1:] == b.shape[dim + 1:], (a.shape, b.shape, dim)
    outer = 1
    for s_ in b.shape[:dim]:
        outer *= s_
    inner = 1
    for s_ in b.shape[dim + 1:]:
        inner *= s_
    _check(lib.osk_blend_bf16(a.data_ptr(), b.data_ptr(), outer, a.shape[dim], b.shape[dim], extent, inner, _stream()),
           "osk_blend_bf16")
    return b


_HD512_WS: dict = {}


def attention_hd512_workspace(B: int, S: int, device) -> torch.Tensor:
    """workspace of the key-split launch, one per (device, stream, size); reused by every call"""
    dev = torch.device(device)
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0, B, S)
    ws = _HD512_WS.get(key)
    if ws is None:
        if len(_HD512_WS) >= 4:
            _HD512_WS.pop(next(iter(_HD512_WS)))
        ws = _HD512_WS[key] = torch.empty(int(lib.osk_attention_hd512_workspace_bytes(B, S)), dtype=torch.uint8, device=dev)
    return ws


def attention_hd512(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, bias_v, out: torch.Tensor, keys_per_frame: int,
                    scale: float, workspace: torch.Tensor | None = None) -> torch.Tensor:
    """the VAE mid block's one-head attention: q, k, out bf16 [B, S, 512] views; vt bf16 [B, 512, ld] (natural key order,
    zero beyond S, ld >= round_up(S, 32)); bias_v f32 [512] | None; frame-causal over groups of keys_per_frame keys."""
    B, S, C = q.shape
    assert C == 512 and vt.shape[1] == 512 and vt.stride(2) == 1 and q.stride(2) == 1 and k.stride(2) == 1 and out.stride(2) == 1
    _check(lib.osk_attention_hd512_fwd_ws_bf16(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1),
                                               vt.data_ptr(), vt.stride(0), vt.stride(1), _p(bias_v), out.data_ptr(),
                                               out.stride(0), out.stride(1), B, S, keys_per_frame, scale, _p(workspace),
                                               0 if workspace is None else workspace.numel() * workspace.element_size(), _stream()),
           "osk_attention_hd512_fwd_ws_bf16")
    return out
