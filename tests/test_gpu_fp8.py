"""GPU parity tests of the opt-in fp8 (OCP e4m3fn) GEMM path (BASELINE configs[4]: "fp8 MFMA"), through the C ABI.

Two separate claims, tested separately:
  * the quantiser is bit-exact against torch's own float8_e4m3fn conversion on the CPU (same f32 arithmetic);
  * the fp8 GEMM kernel equals an f64 matmul of the DEQUANTISED operands up to f32 accumulation + one bf16 rounding
    -- i.e. the kernel adds no error of its own; the quantisation error itself (vs the bf16 path, which is what the
    reference computes) is measured and bounded in test_fp8_error_vs_bf16_path.
"""
import pytest
import torch

from tests.test_gpu_kernels import BF, DEV, bf16_ulp_close, rnd

pytestmark = pytest.mark.gpu
F8 = torch.float8_e4m3fn


def _quant_ref(x):
    """x bf16 [M, K] on the CPU -> (uint8 e4m3 bytes, f32 scales): the arithmetic of osk_quantize_rows_fp8 in f32"""
    xf = x.float()
    amax = xf.abs().amax(-1)
    inv = torch.where(amax > 0, torch.tensor(448.0) / amax, torch.zeros_like(amax))
    q = (xf * inv[:, None]).clamp(-448.0, 448.0).to(F8).view(torch.uint8)
    return q, torch.where(amax > 0, amax / torch.tensor(448.0), torch.ones_like(amax))


def _deq(q8, s):
    return q8.cpu().view(F8).double() * s.cpu().double()[:, None]


@pytest.mark.parametrize("M_L,K", [((1, 7), 64), ((2, 130), 1152), ((1, 33), 4608), ((1, 5), 15360)])
def test_quantize_rows_bit_exact(hip_lib, M_L, K):
    B, L = M_L
    x = rnd("x", (B, L, K), std=1.7, seed=61)
    x[0, 0] = 0                      # an all-zero row: scale 1, bytes 0
    x[0, 1, 3] = 300.0               # an outlier that owns the row's scale
    q, s = hip_lib.quantize_rows_fp8(x)
    qr, sr = _quant_ref(x.cpu().view(B * L, K))
    assert torch.equal(s.cpu(), sr)
    # -0 and +0 are the same value: compare after mapping 0x80 -> 0x00
    fix = lambda t: torch.where(t == 0x80, torch.zeros_like(t), t)
    assert torch.equal(fix(q.cpu()), fix(qr))


def test_quantize_rows_strided_view(hip_lib):
    B, L, D = 2, 40, 256
    y = rnd("y", (B, L, 3 * D), seed=62)
    q, s = hip_lib.quantize_rows_fp8(y[:, :, D: 2 * D])
    qr, sr = _quant_ref(y[:, :, D: 2 * D].cpu().reshape(B * L, D))
    assert torch.equal(s.cpu(), sr) and torch.equal(q.cpu() & 0x7F | (q.cpu() & 0x80) * (q.cpu() != 0x80), qr & 0x7F | (qr & 0x80) * (qr != 0x80))


def _fp8_case(hip_lib, B, L, N, K, gelu_from=None, gated=False, out_f32=False, seed=70):
    a = rnd("a", (B, L, K), seed=seed)
    w = rnd("w", (N, K), std=K ** -0.5, seed=seed + 1)
    bias = rnd("b", (N,), std=0.1, dtype=torch.float32, seed=seed + 2)
    a8, sa = hip_lib.quantize_rows_fp8(a)
    w8, sw = hip_lib.quantize_rows_fp8(w)
    out = torch.empty(B, L, N, dtype=torch.float32 if out_f32 else BF, device=DEV)
    res = gate = None
    if gated:
        res = rnd("r", (B, L, N), seed=seed + 3)
        gate = rnd("g", (B, N), std=0.5, dtype=torch.float32, seed=seed + 4)
        out = res.clone()           # res may alias C
        hip_lib.gemm_fp8(a8, sa, w8, sw, bias, out, res=out, gate=gate, gate_batch_stride=gate.stride(0))
    else:
        hip_lib.gemm_fp8(a8, sa, w8, sw, bias, out, gelu_from=gelu_from)
    v = _deq(a8, sa) @ _deq(w8, sw).T + bias.double().cpu()
    v = v.view(B, L, N)
    if gelu_from is not None:
        g = torch.nn.functional.gelu(v[..., gelu_from:].float(), approximate="tanh").double()
        v = torch.cat([v[..., :gelu_from], g], -1)
    if gated:
        v = res.double().cpu() + gate.double().cpu()[:, None] * v
    if out_f32:
        assert (out.cpu().double() - v).abs().max().item() <= 2e-4 * max(1.0, v.abs().max().item())
    else:
        bf16_ulp_close(out.float().cpu(), v.float().bfloat16().float(), rel=2 ** -7, abs_=2e-3)
    return a, w, bias, out


# exact tiles, ragged M and N tails, tiles straddling a batch boundary, one K step (K = 128), both tile widths
@pytest.mark.parametrize("B,L,N,K", [(1, 256, 256, 128), (1, 512, 512, 256), (2, 300, 384, 256), (3, 700, 1152, 1152),
                                     (1, 1000, 520, 256), (2, 1024, 2304, 640), (1, 257, 132, 128), (1, 300, 200, 384)])
def test_gemm_fp8_vs_dequantised_f64(hip_lib, B, L, N, K):
    _fp8_case(hip_lib, B, L, N, K)


def test_gemm_fp8_epilogues(hip_lib):
    _fp8_case(hip_lib, 2, 384, 1024, 256, gelu_from=256, seed=80)       # linear1-style: GELU on the MLP columns only
    _fp8_case(hip_lib, 2, 384, 512, 512, gated=True, seed=81)            # gate * x + residual, in place
    _fp8_case(hip_lib, 1, 300, 260, 256, gated=True, seed=82)            # the same on ragged tiles
    _fp8_case(hip_lib, 1, 512, 256, 256, out_f32=True, seed=83)


def test_gemm_fp8_unsupported_shapes_are_refused(hip_lib):
    a8 = torch.zeros(64, 128, dtype=torch.uint8, device=DEV)
    w8 = torch.zeros(128, 128, dtype=torch.uint8, device=DEV)
    s = torch.ones(128, dtype=torch.float32, device=DEV)
    out = torch.empty(1, 64, 128, dtype=BF, device=DEV)
    with pytest.raises(RuntimeError):                                    # M < 256: this layer stays on the bf16 GEMM
        hip_lib.gemm_fp8(a8, s, w8, s, None, out)
    assert not hip_lib.gemm_fp8_supported(64, 128, 128) and hip_lib.gemm_fp8_supported(256, 128, 128)
    assert not hip_lib.gemm_fp8_supported(256, 128, 192)


def test_fp8_error_vs_bf16_path(hip_lib):
    """quantisation error of one Linear at the XL width against the bf16 kernel (what the reference computes):
    relL2 of the fp8 result vs the f64 product of the bf16 operands; e4m3 with per-row scales gives ~2.5-3 %."""
    B, L, N, K = 1, 2048, 1152, 1152
    a, w, bias, out8 = _fp8_case(hip_lib, B, L, N, K, seed=90)
    ref = a.double().cpu() @ w.double().cpu().T + bias.double().cpu()
    out16 = torch.empty(B, L, N, dtype=BF, device=DEV)
    hip_lib.gemm(a, w, bias, out16)
    e8 = ((out8.double().cpu() - ref).norm() / ref.norm()).item()
    e16 = ((out16.double().cpu() - ref).norm() / ref.norm()).item()
    assert e16 <= 4e-3 and e8 <= 4e-2, (e8, e16)


def test_gemm_fp8_deterministic_under_load(hip_lib):
    a = rnd("a", (3, 2100, 1152), seed=95)
    w = rnd("w", (1160, 1152), std=1152 ** -0.5, seed=96)
    a8, sa = hip_lib.quantize_rows_fp8(a)
    w8, sw = hip_lib.quantize_rows_fp8(w)
    outs = []
    for _ in range(12):
        o = torch.empty(3, 2100, 1160, dtype=BF, device=DEV)
        hip_lib.gemm_fp8(a8, sa, w8, sw, None, o)
        outs.append(o)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])


def test_mmdit_forward_fp8_mode(hip_lib):
    """MMDiTModel.enable_fp8(): the block Linears on the fp8 MFMA.  Gate of SURVEY.md 8(d) for the fp8 configuration:
    relL2 <= 5e-2 against our bf16 path and against the fp32 oracle; the bf16 mode is restored bit-exactly."""
    from oracle import configs, mmdit_oracle as O
    from open_sora_amd import mmdit
    from tests.util import torch_inputs, torch_params

    cfg = dict(configs.GOLDEN["hd128_eager_fused"][0], depth=1, depth_single_blocks=1)
    geom = (2, 2, 12, 12, 160)
    model = mmdit.Flux(device_map="cuda", torch_dtype=BF, **cfg)
    model.load_state_dict(torch_params(cfg, dtype=BF, device="cuda"), strict=True)
    inp = torch_inputs(cfg, *geom, dtype=BF, device="cuda")
    calls = []
    real = hip_lib.gemm_fp8
    try:
        hip_lib.gemm_fp8 = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        with torch.inference_mode():
            ref16 = model(**inp).clone()
            assert not calls
            out8 = model.enable_fp8()(**inp).clone()
            n8 = len(calls)
            again16 = model.enable_fp8(False)(**inp).clone()
    finally:
        hip_lib.gemm_fp8 = real
    assert n8 == 10 and len(calls) == 10
    assert torch.equal(again16, ref16)
    with torch.inference_mode():
        truth = O.forward(torch_params(cfg), cfg, **torch_inputs(cfg, *geom))
    rel = lambda a, b: ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm()).item()
    assert rel(out8, ref16) <= 5e-2 and rel(out8, truth) <= 5e-2, (rel(out8, ref16), rel(out8, truth))


# ----------------------------------------------------------------------------- attention with the fp8 P.V product
def _pv8_case(hip_lib, B, H, hd, Lq, Lk, n_seg=1, spike=False, workspace=False, seed=120):
    D = H * hd
    seg = Lk // n_seg
    q = rnd("q", (B, Lq, D), seed=seed)
    k = rnd("k", (n_seg, B, seg, D), seed=seed + 1)
    v = rnd("v", (n_seg, B, seg, D), seed=seed + 2)
    if spike:
        k[-1, :, seg - 3] = q[:, 0] * 4.0
    segp = (seg + 63) // 64 * 64
    RP = hip_lib.vt8_rows(hd)
    # one scale per (batch, head) over ALL segments (the kernel accumulates across segments)
    amax = v.float().abs().view(n_seg, B, seg, H, hd).amax(dim=(0, 2, 4))
    sv = (amax / 448.0).contiguous()
    vt8 = torch.empty(n_seg, B, H, RP, segp, dtype=torch.uint8, device=DEV)
    for s_ in range(n_seg):
        hip_lib.v_transpose_fp8(v[s_], sv, vt8[s_], H, hd)
    out = torch.empty(B, Lq, D, dtype=BF, device=DEV)
    lse = torch.empty(B, H, Lq, dtype=torch.float32, device=DEV)
    ws = hip_lib.attention_workspace(q.device) if workspace else None
    hip_lib.attention_fwd_pv8(q, k[0], vt8, sv, out, H, hd, hd ** -0.5, lse=lse, n_seg=n_seg, seg_len=seg,
                              k_seg_stride=k.stride(0), vt_seg_stride=vt8.stride(0), workspace=ws)
    kk = k.permute(1, 0, 2, 3).reshape(B, n_seg * seg, H, hd).double().permute(0, 2, 1, 3)
    vv = v.permute(1, 0, 2, 3).reshape(B, n_seg * seg, H, hd).float()
    v8 = ((vv / sv[:, None, :, None]).clamp(-448, 448).to(F8).float() * sv[:, None, :, None]).double().permute(0, 2, 1, 3)
    qh = q.double().view(B, Lq, H, hd).permute(0, 2, 1, 3)
    s_ = (qh @ kk.transpose(-1, -2)) * hd ** -0.5
    p_ = torch.softmax(s_, -1)
    ref8 = (p_ @ v8).permute(0, 2, 1, 3).reshape(B, Lq, D)                 # exact P, the kernel's e4m3 V
    ref = (p_ @ vv.double().permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(B, Lq, D)
    o = out.double()
    rel8 = ((o - ref8).norm() / ref8.norm()).item()
    rel = ((o - ref).norm() / ref.norm()).item()
    # e4m3 has 3 mantissa bits: P and V each carry ~3 % rounding noise per element
    assert rel8 <= 4e-2 and rel <= 6e-2, (rel8, rel)
    # the softmax denominator is the sum of the e4m3 P: a row dominated by one key carries that key's rounding,
    # up to half an e4m3 step (2^-4): ln(1 + 2^-4) = 0.061
    assert (lse.double() - torch.logsumexp(s_, -1)).abs().max().item() <= 7e-2
    return out


@pytest.mark.parametrize("hd", [72, 128])
@pytest.mark.parametrize("Lq,Lk", [(256, 256), (300, 1000), (64, 65), (512, 4096), (33, 704)])
def test_attention_pv8_vs_f64(hip_lib, hd, Lq, Lk):
    _pv8_case(hip_lib, 2, 2, hd, Lq, Lk)


@pytest.mark.parametrize("hd", [72, 128])
def test_attention_pv8_segments_spike_and_tail_split(hip_lib, hd):
    _pv8_case(hip_lib, 2, 2, hd, 130, 300, n_seg=3, seed=130)               # ragged key segments
    _pv8_case(hip_lib, 1, 2, hd, 128, 900, spike=True, seed=131)            # reference-max move (rare path)
    _pv8_case(hip_lib, 1, 17, hd, 4096, 1000, workspace=True, seed=132)     # tail units split along the keys
    _pv8_case(hip_lib, 2, 9, hd, 4000, 400, n_seg=4, workspace=True, seed=133)


@pytest.mark.parametrize("hd", [72, 128])
def test_attention_pv8_constant_v_and_determinism(hip_lib, hd):
    """V == c per channel: out == e4m3(c / s) * s exactly (the denominator is the ones row of the SAME fp8 product);
    12 runs bit-identical."""
    B, H, Lq, Lk = 1, 8, 2000, 4133
    D = H * hd
    q, k = rnd("q", (B, Lq, D), seed=141), rnd("k", (B, Lk, D), seed=142)
    c = rnd("c", (D,), seed=143)
    v = c[None, None].expand(B, Lk, D).contiguous()
    sv = (v.float().abs().view(B, Lk, H, hd).amax(dim=(1, 3)) / 448.0).contiguous()
    vt8 = torch.empty(B, H, hip_lib.vt8_rows(hd), (Lk + 63) // 64 * 64, dtype=torch.uint8, device=DEV)
    hip_lib.v_transpose_fp8(v, sv, vt8, H, hd)
    outs = []
    for _ in range(12):
        o = torch.empty(B, Lq, D, dtype=BF, device=DEV)
        hip_lib.attention_fwd_pv8(q, k, vt8, sv, o, H, hd, hd ** -0.5)
        outs.append(o)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    c8 = (c.float().view(H, hd) / sv[0][:, None]).clamp(-448, 448).to(F8).float() * sv[0][:, None]
    assert (outs[0].float() - c8.view(1, 1, D)).abs().max().item() <= 2 ** -7 * c8.abs().max().item() + 1e-3


@pytest.mark.parametrize("D", [256, 1152, 3072])
def test_ln_modulate_fp8_equals_the_two_kernel_path(hip_lib, D):
    """osk_ln_modulate_fp8 == osk_ln_modulate_bf16 + osk_quantize_rows_fp8, bit for bit (bytes and row scales),
    on a strided input view."""
    B, L = 2, 133
    buf = rnd("x", (B, L + 5, D), std=2.0, seed=151)
    x = buf[:, 5:]
    mod = rnd("mod", (B, 2 * D + 8), std=0.5, dtype=torch.float32, seed=152)
    shift, scale = mod[:, :D], mod[:, D + 8: 2 * D + 8]
    xm = torch.empty(B, L, D, dtype=BF, device=DEV)
    hip_lib.ln_modulate(x, shift, scale, xm, mod.stride(0))
    q_ref, s_ref = hip_lib.quantize_rows_fp8(xm)
    q, s = hip_lib.ln_modulate_fp8(x, shift, scale, mod.stride(0))
    assert torch.equal(s, s_ref) and torch.equal(q, q_ref)


@pytest.mark.parametrize("H,hd,L", [(16, 72, 1000), (24, 128, 333), (2, 64, 77)])
def test_v_scale_fp8_exact(hip_lib, H, hd, L):
    B = 3
    y = rnd("y", (B, L, 3 * H * hd), seed=161)
    v = y[:, :, 2 * H * hd:]
    v[1, :, hd: 2 * hd] = 0            # an all-zero head: scale 1
    s = hip_lib.v_scale_fp8(v, H, hd)
    amax = v.float().view(B, L, H, hd).abs().amax(dim=(1, 3)).cpu()     # the division on the CPU: IEEE, like the kernel's
    ref = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    assert torch.equal(s.cpu(), ref)
