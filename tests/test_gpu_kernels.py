"""GPU parity tests, kernel by kernel: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Tolerances are stated per test; bf16 outputs are compared after the same final rounding."""
import math

import pytest
import torch

from oracle import mmdit_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu

DEV = "cuda"
BF = torch.bfloat16


def rnd(name, shape, std=1.0, seed=7, dtype=BF):
    return torch.from_numpy(synth.normal(name, seed, shape, std=std)).to(DEV).to(dtype)


def bf16_ulp_close(a, b, rel=2 ** -7, abs_=1e-3):
    """a, b float tensors; allow one bf16 rounding step of disagreement."""
    d = (a.double() - b.double()).abs()
    tol = abs_ + rel * b.double().abs()
    bad = (d > tol).sum().item()
    assert bad == 0, f"{bad} elements differ; max err {d.max().item():.4e}"


# ----------------------------------------------------------------------------- LN + modulate
@pytest.mark.parametrize("D", [128, 384, 1152, 3072])
def test_ln_modulate(hip_lib, D):
    B, L = 2, 37
    x = rnd("x", (B, L, D), std=2.0)
    mod = rnd("mod", (B, 2 * D + 8), std=0.5, dtype=torch.float32)
    shift, scale = mod[:, :D], mod[:, D + 8: 2 * D + 8]
    out = torch.empty_like(x)
    hip_lib.ln_modulate(x, shift, scale, out, mod.stride(0))
    xf = x.float().cpu()
    ref = (1 + scale.cpu()[:, None]) * O._layer_norm(xf) + shift.cpu()[:, None]
    bf16_ulp_close(out.float().cpu(), ref.bfloat16().float())


def test_ln_modulate_strided_views(hip_lib):
    B, Lt, Li, D = 2, 5, 11, 256
    buf = rnd("x", (B, Lt + Li, D))
    outb = torch.zeros_like(buf)
    mod = rnd("mod", (B, 2 * D), std=0.3, dtype=torch.float32)
    hip_lib.ln_modulate(buf[:, Lt:], mod[:, :D], mod[:, D:], outb[:, Lt:], mod.stride(0))
    ref = (1 + mod.cpu()[:, None, D:]) * O._layer_norm(buf[:, Lt:].float().cpu()) + mod.cpu()[:, None, :D]
    bf16_ulp_close(outb[:, Lt:].float().cpu(), ref.bfloat16().float())
    assert outb[:, :Lt].abs().max().item() == 0.0  # nothing written outside the view


# ----------------------------------------------------------------------------- GEMM
def _gemm_ref(a, w, bias, gelu_from=None, res=None, gate=None):
    v = a.double().cpu() @ w.double().cpu().T
    if bias is not None:
        v = v + bias.double().cpu()
    if gelu_from is not None:
        g = torch.nn.functional.gelu(v[..., gelu_from:].float(), approximate="tanh").double()
        v = torch.cat([v[..., :gelu_from], g], -1)
    if gate is not None:
        v = res.double().cpu() + gate.double().cpu()[:, None] * v
    return v


@pytest.mark.parametrize("M_L,N,K", [((1, 128), 128, 64), ((2, 200), 384, 128), ((1, 77), 64, 192), ((3, 130), 1152, 1152), ((1, 513), 200, 256)])
def test_gemm_plain_bias(hip_lib, M_L, N, K):
    B, L = M_L
    a = rnd("a", (B, L, K))
    w = rnd("w", (N, K), std=K ** -0.5)
    bias = rnd("b", (N,), std=0.1, dtype=torch.float32)
    out = torch.empty(B, L, N, dtype=BF, device=DEV)
    hip_lib.gemm(a, w, bias, out)
    ref = _gemm_ref(a, w, bias)
    bf16_ulp_close(out.float().cpu(), ref.float().bfloat16().float(), rel=2 ** -7, abs_=2e-3)
    out32 = torch.empty(B, L, N, dtype=torch.float32, device=DEV)
    hip_lib.gemm(a, w, None, out32)
    ref32 = _gemm_ref(a, w, None)
    assert (out32.cpu().double() - ref32).abs().max().item() <= 1e-3 * max(1.0, ref32.abs().max().item())


# shapes that dispatch to the large-tile hand-scheduled kernel (gemm256.hip: M >= 256, N >= 128): exact tiles, ragged M
# and N tails, tiles straddling a batch boundary, one K step (K = 64), both tile widths (N % 256 == 0 -> BN 256, else 128)
@pytest.mark.parametrize("M_L,N,K", [((1, 256), 256, 64), ((1, 512), 512, 128), ((2, 300), 384, 192), ((3, 700), 1152, 1152),
                                     ((1, 1000), 520, 256), ((2, 1024), 2304, 576), ((1, 257), 132, 64), ((1, 300), 200, 128),
                                     # the 11B geometry's Linear shapes (hidden 3072: QKV, linear1 = QKV + MLP-up, MLP-down, linear2):
                                     # K = 3072 / 12288 / 15360 are 48 / 192 / 240 K steps of the persistent loop, N = 21,504 is 84 column tiles
                                     ((3, 1300), 9216, 3072), ((1, 2560), 21504, 3072), ((2, 700), 3072, 12288), ((1, 1300), 3072, 15360)])
def test_gemm_large_tile(hip_lib, M_L, N, K):
    B, L = M_L
    a = rnd("a", (B, L, K))
    w = rnd("w", (N, K), std=K ** -0.5)
    bias = rnd("b", (N,), std=0.1, dtype=torch.float32)
    out = torch.empty(B, L, N, dtype=BF, device=DEV)
    hip_lib.gemm(a, w, bias, out)
    bf16_ulp_close(out.float().cpu(), _gemm_ref(a, w, bias).float().bfloat16().float(), rel=2 ** -7, abs_=2e-3)
    out32 = torch.empty(B, L, N, dtype=torch.float32, device=DEV)
    hip_lib.gemm(a, w, None, out32)
    ref32 = _gemm_ref(a, w, None)
    assert (out32.cpu().double() - ref32).abs().max().item() <= 1e-3 * max(1.0, ref32.abs().max().item())


def test_gemm_large_tile_epilogues_and_views(hip_lib):
    """single-block shaped at large-tile sizes: linear1 (GELU from col 3D) into a wide buffer, then linear2 reading the
    [attn | gelu(mlp)] column view with gate * x + residual written in place (batch rows = 2 tiles + a tail)."""
    B, L, D, R = 2, 600, 256, 1024
    x = rnd("x", (B, L, D))
    w1 = rnd("w1", (3 * D + R, D), std=D ** -0.5)
    b1 = rnd("b1", (3 * D + R,), std=0.1, dtype=torch.float32)
    y = torch.empty(B, L, 3 * D + R, dtype=BF, device=DEV)
    hip_lib.gemm(x, w1, b1, y, gelu_from=3 * D)
    bf16_ulp_close(y.float().cpu(), _gemm_ref(x, w1, b1, gelu_from=3 * D).float().bfloat16().float(), abs_=3e-3)
    w2 = rnd("w2", (D, D + R), std=(D + R) ** -0.5)
    b2 = rnd("b2", (D,), std=0.1, dtype=torch.float32)
    gate = rnd("g", (B, 3 * D), std=0.5, dtype=torch.float32)
    res = x.clone()
    hip_lib.gemm(y[:, :, 2 * D:], w2, b2, res, res=res, gate=gate[:, 2 * D:], gate_batch_stride=gate.stride(0))
    ref2 = _gemm_ref(y[:, :, 2 * D:], w2, b2, res=x, gate=gate[:, 2 * D:])
    bf16_ulp_close(res.float().cpu(), ref2.float().bfloat16().float(), abs_=3e-3)
    # joint-buffer rows: two GEMMs write disjoint row ranges of one buffer
    Lt = 280
    yj = torch.zeros(B, L, 3 * D, dtype=BF, device=DEV)
    wi, wt = rnd("wi", (3 * D, D), std=D ** -0.5), rnd("wt", (3 * D, D), std=D ** -0.5)
    hip_lib.gemm(x[:, Lt:], wi, None, yj[:, Lt:])
    hip_lib.gemm(x[:, :Lt], wt, None, yj[:, :Lt])
    bf16_ulp_close(yj[:, Lt:].float().cpu(), _gemm_ref(x[:, Lt:], wi, None).float().bfloat16().float(), abs_=3e-3)
    bf16_ulp_close(yj[:, :Lt].float().cpu(), _gemm_ref(x[:, :Lt], wt, None).float().bfloat16().float(), abs_=3e-3)


# The persistent-workgroup kernel (gemm256p.hip) only differs from a one-tile-per-workgroup launch when a workgroup owns
# SEVERAL tiles (more tiles than CUs): cross-tile LDS-DMA prefetch at the end of a K loop, prefetched entry, bias-initialised
# accumulators per tile.  Shapes with 2-5 tiles per workgroup: one / two / odd K-step counts, ragged M and N edges, batch
# boundaries inside tiles, GELU starting inside a 32-column tile, the gate epilogue, f32 output, both tile widths.
@pytest.mark.parametrize("M_L,N,K,gelu_from,gated", [
    ((3, 5000), 2304, 192, 1000, False),     # 59 x 9 = 531 tiles; GELU from a column that is not a multiple of 32
    ((3, 9000), 1152, 256, None, True),      # 106 x 5 tiles (last N tile half empty), gate * x + residual
    ((1, 40000), 512, 64, None, False),      # one K step per tile (157 x 2 tiles)
    ((2, 20000), 640, 128, 384, False),      # two K steps, BN = 128 path (N % 256 != 0), GELU from a tile boundary
    ((1, 33000), 1000, 320, None, True),     # ragged N (1000), ragged M, odd K-step count, gated
])
def test_gemm_persistent_multi_tile(hip_lib, M_L, N, K, gelu_from, gated):
    B, L = M_L
    a = rnd("a", (B, L, K), seed=31)
    w = rnd("w", (N, K), std=K ** -0.5, seed=32)
    bias = rnd("b", (N,), std=0.3, dtype=torch.float32, seed=33)
    res = gate = None
    out = torch.empty(B, L, N, dtype=BF, device=DEV)
    kw = {}
    if gated:
        res = rnd("r", (B, L, N), seed=34)
        gate = rnd("g", (B, N), std=0.5, dtype=torch.float32, seed=35)
        out.copy_(res)
        kw = dict(res=out, gate=gate, gate_batch_stride=gate.stride(0))
    hip_lib.gemm(a, w, bias, out, gelu_from=gelu_from, **kw)
    # reference on the GPU in f32 (operands are bf16-exact, so f32 accumulation is the only difference from f64)
    v = (a.float().reshape(B * L, K) @ w.float().T + bias).reshape(B, L, N)
    if gelu_from is not None:
        v = torch.cat([v[..., :gelu_from], torch.nn.functional.gelu(v[..., gelu_from:], approximate="tanh")], -1)
    if gated:
        v = res.float() + gate[:, None, :] * v
    bf16_ulp_close(out.float().cpu(), v.bfloat16().float().cpu(), rel=2 ** -7, abs_=3e-3)
    outs = []
    for _ in range(3):
        o = torch.empty(B, L, N, dtype=BF, device=DEV)
        if gated:
            o.copy_(res)
            hip_lib.gemm(a, w, bias, o, gelu_from=gelu_from, res=o, gate=gate, gate_batch_stride=gate.stride(0))
        else:
            hip_lib.gemm(a, w, bias, o, gelu_from=gelu_from)
        outs.append(o)
    torch.cuda.synchronize()
    assert all(torch.equal(out, o) for o in outs), "persistent GEMM: run-to-run difference (cross-tile prefetch race?)"
    if not gated:
        o32 = torch.empty(B, L, N, dtype=torch.float32, device=DEV)
        hip_lib.gemm(a, w, None, o32)
        ref32 = (a.float().reshape(B * L, K) @ w.float().T).reshape(B, L, N)
        assert (o32 - ref32).abs().max().item() <= 1e-3 * max(1.0, ref32.abs().max().item())


@pytest.mark.parametrize("kind", [0, 1, 2])       # 128 x 128 (gemm_bf16_kernel), 256 x 128 (gemm256p_kernel), 256 x 256 (gemm256x_kernel)
@pytest.mark.parametrize("gelu_from,gated", [(None, True), (640, False)])
def test_gemm_every_tile_kernel_on_the_same_problem(hip_lib, kind, gelu_from, gated):
    """Which tile kernel serves a shape is the dispatcher's estimate (osk_gemm_tile_choice; it moved in round 5), so coverage of a
    kernel's epilogue classes by shape is coverage by luck.  Here every tile kernel is forced (osk_gemm_tile_override) over the same
    problem -- the gate * x + residual class written in place (each kernel reads its residual in its own layout) and a GELU boundary
    inside a tile -- against the f32 reference; ragged M (batch boundary inside a tile) and ragged N for the 256-wide tiles."""
    B, L, N, K = 2, 3000, 1152, 256
    a = rnd("a", (B, L, K), seed=51)
    w = rnd("w", (N, K), std=K ** -0.5, seed=52)
    bias = rnd("b", (N,), std=0.3, dtype=torch.float32, seed=53)
    out = torch.empty(B, L, N, dtype=BF, device=DEV)
    kw = {}
    if gated:
        res = rnd("r", (B, L, N), seed=54)
        gate = rnd("g", (B, N), std=0.5, dtype=torch.float32, seed=55)
        out.copy_(res)
        kw = dict(res=out, gate=gate, gate_batch_stride=gate.stride(0))
    assert hip_lib.lib.osk_gemm_tile_override(kind) == 0
    try:
        hip_lib.gemm(a, w, bias, out, gelu_from=gelu_from, **kw)
        torch.cuda.synchronize()
    finally:
        hip_lib.lib.osk_gemm_tile_override(-1)
    v = (a.float().reshape(B * L, K) @ w.float().T + bias).reshape(B, L, N)
    if gelu_from is not None:
        v = torch.cat([v[..., :gelu_from], torch.nn.functional.gelu(v[..., gelu_from:], approximate="tanh")], -1)
    if gated:
        v = res.float() + gate[:, None, :] * v
    bf16_ulp_close(out.float().cpu(), v.bfloat16().float().cpu(), rel=2 ** -7, abs_=3e-3)


@pytest.mark.parametrize("B,Li,Lt,N,K,gelu_from,gated", [
    (3, 4000, 512, 3456, 1152, None, False),     # double-block QKV at a reduced token count: both problems on the 256 x 256 tiles
    (3, 8000, 300, 1152, 1152, None, True),      # proj: gate * x + residual written in place, ragged N (4.5 column tiles), ragged M
    (2, 8000, 512, 1024, 256, 0, False),         # MLP-up kind: GELU over the whole row
    (1, 200, 77, 384, 128, None, True),          # small: the entry falls back to the two single calls (bit-identical to them)
    (2, 8192, 512, 3072, 3072, None, True),      # 11B width: the image problem's 768 tiles are exactly 3 rounds -> the entry declines (two calls)
    (1, 6144, 512, 9216, 3072, None, False),     # 11B QKV: 24 x 36 + 2 x 36 tiles, K = 3072
])
def test_gemm_pair_equals_two_calls(hip_lib, B, Li, Lt, N, K, gelu_from, gated):
    """osk_gemm_bf16_pair on the joint-buffer views of a double block: [txt ; img] rows of one buffer, separate weights."""
    L = Lt + Li
    x = rnd("x", (B, L, K), seed=41)
    w = [rnd("wi", (N, K), std=K ** -0.5, seed=42), rnd("wt", (N, K), std=K ** -0.5, seed=43)]
    b = [rnd("bi", (N,), std=0.2, dtype=torch.float32, seed=44), rnd("bt", (N,), std=0.2, dtype=torch.float32, seed=45)]
    gate = [rnd("gi", (B, N), std=0.5, dtype=torch.float32, seed=46), rnd("gt", (B, N), std=0.5, dtype=torch.float32, seed=47)]
    res0 = rnd("r", (B, L, N), seed=48)
    views = lambda t: (t[:, Lt:], t[:, :Lt])          # img rows, txt rows

    def run(pair):
        out = res0.clone() if gated else torch.full((B, L, N), float("nan"), dtype=BF, device=DEV)
        args = []
        for i, (a, o) in enumerate(zip(views(x), views(out))):
            d = dict(a=a, w=w[i], bias=b[i], out=o)
            if gated:
                d.update(res=o, gate=gate[i], gate_batch_stride=gate[i].stride(0))
            args.append(d)
        if pair:
            hip_lib.gemm_pair(args[0], args[1], gelu_from=gelu_from)
        else:
            for d in args:
                hip_lib.gemm(d["a"], d["w"], d["bias"], d["out"], gelu_from=gelu_from,
                             **{k: d[k] for k in ("res", "gate", "gate_batch_stride") if k in d})
        return out

    got, single = run(True), run(False)
    for i, (a, o) in enumerate(zip(views(x), views(got))):
        ref = _gemm_ref(a, w[i], b[i], gelu_from=gelu_from, res=views(res0)[i] if gated else None, gate=gate[i] if gated else None)
        bf16_ulp_close(o.float().cpu(), ref.float().bfloat16().float(), rel=2 ** -7, abs_=3e-3)
    # against the single calls: the image problem runs the kernel it would run alone (bit-identical); the text problem may move
    # from the 128-wide tiles to this launch's 256-wide ones (another accumulation order)
    assert torch.equal(views(got)[0], views(single)[0])
    bf16_ulp_close(got.float().cpu(), single.float().cpu(), rel=2 ** -7, abs_=3e-3)
    if B * L < 1024:
        assert torch.equal(got, single)
    again = run(True)
    assert torch.equal(got, again), "pair launch is not repeatable"


def test_gemm_is_transpose_detecting(hip_lib):
    """A = I block, asymmetric W: C must equal W^T rows exactly (guide: A=I-check with asymmetric B)."""
    K = N = 128
    a = torch.eye(K, dtype=BF, device=DEV)[None]
    w = (torch.arange(N * K, device=DEV).reshape(N, K) % 251).to(BF)
    out = torch.empty(1, K, N, dtype=torch.float32, device=DEV)
    hip_lib.gemm(a, w, None, out)
    assert torch.equal(out[0].cpu(), w.float().cpu().T)


def test_gemm_gelu_gate_residual_views(hip_lib):
    """single-block shaped: linear1 (GELU from col 3D) then linear2 reading [attn|gelu(mlp)] columns in place."""
    B, L, D, R = 2, 150, 128, 512
    x = rnd("x", (B, L, D))
    w1 = rnd("w1", (3 * D + R, D), std=D ** -0.5)
    b1 = rnd("b1", (3 * D + R,), std=0.1, dtype=torch.float32)
    y = torch.empty(B, L, 3 * D + R, dtype=BF, device=DEV)
    hip_lib.gemm(x, w1, b1, y, gelu_from=3 * D)
    ref1 = _gemm_ref(x, w1, b1, gelu_from=3 * D)
    bf16_ulp_close(y.float().cpu(), ref1.float().bfloat16().float(), abs_=3e-3)
    w2 = rnd("w2", (D, D + R), std=(D + R) ** -0.5)
    b2 = rnd("b2", (D,), std=0.1, dtype=torch.float32)
    gate = rnd("g", (B, 3 * D), std=0.5, dtype=torch.float32)
    res = x.clone()
    hip_lib.gemm(y[:, :, 2 * D:], w2, b2, res, res=res, gate=gate[:, 2 * D:], gate_batch_stride=gate.stride(0))
    ref2 = _gemm_ref(y[:, :, 2 * D:], w2, b2, res=x, gate=gate[:, 2 * D:])
    bf16_ulp_close(res.float().cpu(), ref2.float().bfloat16().float(), abs_=3e-3)


def test_gemm_joint_buffer_rows(hip_lib):
    """double-block shaped: txt and img GEMMs write disjoint row ranges of one [B, L, 3D] buffer."""
    B, Lt, Li, D = 2, 40, 90, 128
    y = torch.zeros(B, Lt + Li, 3 * D, dtype=BF, device=DEV)
    xi, xt = rnd("xi", (B, Li, D)), rnd("xt", (B, Lt, D))
    wi, wt = rnd("wi", (3 * D, D), std=D ** -0.5), rnd("wt", (3 * D, D), std=D ** -0.5)
    hip_lib.gemm(xi, wi, None, y[:, Lt:])
    hip_lib.gemm(xt, wt, None, y[:, :Lt])
    bf16_ulp_close(y[:, Lt:].float().cpu(), _gemm_ref(xi, wi, None).float().bfloat16().float(), abs_=3e-3)
    bf16_ulp_close(y[:, :Lt].float().cpu(), _gemm_ref(xt, wt, None).float().bfloat16().float(), abs_=3e-3)


# ----------------------------------------------------------------------------- QK norm + RoPE
@pytest.mark.parametrize("hd,axes", [(64, [16, 24, 24]), (72, [8, 32, 32]), (128, [16, 56, 56])])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("H", [3, 6, 16])   # head_dim 64 / 72: H < 4 takes the lane-group kernel, H >= 4 the row-per-thread kernel (S: 6, XL: 16)
def test_qknorm_rope(hip_lib, hd, axes, mode, H):
    if hd == 128 and H != 3:
        pytest.skip("head_dim 128 has one kernel")
    B, Lt, Li = 2, 9, 70
    L, D = Lt + Li, H * hd
    y = rnd("qkv", (B, L, 3 * D), std=1.5)
    y0 = y.clone()
    scales = [rnd(f"s{i}", (hd,), std=0.2).float().add(1.0).to(BF) for i in range(4)]
    ids = torch.zeros(B, L, 3)
    g = torch.stack(torch.meshgrid(torch.arange(2), torch.arange(5), torch.arange(7), indexing="ij"), -1).reshape(-1, 3).float()
    ids[:, Lt:] = g[None]
    ang = (O.rope_angles_liger if mode == 1 else O.rope_angles)(ids, axes, 10000)
    cos = torch.empty(B, L, hd // 2, dtype=torch.float32, device=DEV)
    sin = torch.empty_like(cos)
    hip_lib.rope_table(ids.view(B * L, 3).to(DEV), axes, 10000, mode == 1, cos, sin)
    assert (cos.cpu() - torch.cos(ang).float()).abs().max() < 2e-5
    assert (sin.cpu() - torch.sin(ang).float()).abs().max() < 2e-5
    q, k = y[:, :, :D], y[:, :, D: 2 * D]
    hip_lib.qknorm_rope(q, k, scales[0], scales[1], scales[2], scales[3], Lt, cos, sin, cos.stride(0), H, hd, mode)
    rope = O.apply_rope_half if mode == 1 else O.apply_rope_interleaved
    for which, (st, sm) in enumerate([(scales[0], scales[2]), (scales[1], scales[3])]):
        x0 = y0[:, :, which * D: (which + 1) * D].cpu().view(B, L, H, hd).permute(0, 2, 1, 3)
        n_t = O.rms_norm(x0[:, :, :Lt], st.cpu())
        n_i = O.rms_norm(x0[:, :, Lt:], sm.cpu())
        ref = rope(torch.cat([n_t, n_i], 2), ang)  # bf16 tensors -> reference rounding points
        got = y[:, :, which * D: (which + 1) * D].cpu().view(B, L, H, hd).permute(0, 2, 1, 3)
        bf16_ulp_close(got.float(), ref.float(), rel=2 ** -7, abs_=2e-3)
    assert torch.equal(y[:, :, 2 * D:], y0[:, :, 2 * D:])  # v untouched
    # q_mult: the softmax scale folded into q before q's single rounding; k is unaffected
    c = hd ** -0.5 * 1.4426950408889634
    y2 = y0.clone()
    hip_lib.qknorm_rope(y2[:, :, :D], y2[:, :, D: 2 * D], scales[0], scales[1], scales[2], scales[3], Lt, cos, sin,
                        cos.stride(0), H, hd, mode, q_mult=c)
    assert torch.equal(y2[:, :, D:], y[:, :, D:])
    x0 = y0[:, :, :D].cpu().view(B, L, H, hd).permute(0, 2, 1, 3)
    nq = torch.cat([O.rms_norm(x0[:, :, :Lt], scales[0].cpu()), O.rms_norm(x0[:, :, Lt:], scales[2].cpu())], 2)
    ref32 = rope(nq.float(), ang) * c          # rotate in f32, scale in f32, ONE rounding
    got = y2[:, :, :D].cpu().view(B, L, H, hd).permute(0, 2, 1, 3)
    bf16_ulp_close(got.float(), ref32.bfloat16().float(), rel=2 ** -7, abs_=1e-3)


# ----------------------------------------------------------------------------- V transpose (bit exact)
@pytest.mark.parametrize("hd", [64, 72, 128])
def test_v_transpose_exact(hip_lib, hd):
    B, L, H = 2, 150, 2
    D = H * hd
    y = rnd("v", (B, L, 3 * D))
    v = y[:, :, 2 * D:]
    Lp = (L + 63) // 64 * 64
    vt = torch.full((B, H, hd, Lp), 7.0, dtype=BF, device=DEV)
    hip_lib.v_transpose(v, vt, H, hd)
    from tests.cpu_ops import pos2key   # the documented key order per head_dim (16x16x32 P.V for 72, 32x32x16 otherwise)

    vpad = torch.zeros(B, Lp, H, hd, dtype=BF)
    vpad[:, :L] = v.cpu().view(B, L, H, hd)
    ref = vpad[:, pos2key(hd, Lp)].permute(0, 2, 3, 1)
    assert torch.equal(vt.cpu(), ref)


# ----------------------------------------------------------------------------- attention
def _attn_case(hip_lib, B, H, hd, Lq, Lk, seed=3, spike=False, lse_tol=2e-3, prescaled=False):
    """prescaled: q already carries scale*log2(e) (what osk_qknorm_rope_bf16's q_mult produces for the model path);
    the fp64 reference is then taken on exactly those bf16 values, in base 2."""
    D = H * hd
    q = rnd("q", (B, Lq, D), seed=seed)
    kv = rnd("kv", (B, Lk, 2 * D), seed=seed + 1)
    if spike:  # force a large running-max jump late in the key sequence (online-softmax rescale path)
        kv[:, Lk - 3, :D] = q[:, 0, :] * 4.0
    if prescaled:
        q = (q.float() * (hd ** -0.5 * 1.4426950408889634)).to(BF)
    k, v = kv[:, :, :D], kv[:, :, D:]
    Lp = (Lk + 63) // 64 * 64
    vt = torch.empty(B, H, hd, Lp, dtype=BF, device=DEV)
    hip_lib.v_transpose(v, vt, H, hd)
    out = torch.empty(B, Lq, D, dtype=BF, device=DEV)
    lse = torch.empty(B, H, Lq, dtype=torch.float32, device=DEV)
    hip_lib.attention_fwd(q, k, vt, out, H, hd, hd ** -0.5, lse=lse, q_prescaled=prescaled)

    def heads(t, L):
        return t.float().cpu().view(B, L, H, hd).permute(0, 2, 1, 3)

    qh, kh, vh = heads(q, Lq), heads(k, Lk), heads(v, Lk)
    s = (qh.double() @ kh.double().transpose(-1, -2)) * (0.6931471805599453 if prescaled else hd ** -0.5)
    ref = (torch.softmax(s, -1) @ vh.double()).permute(0, 2, 1, 3).reshape(B, Lq, D)
    ref_lse = torch.logsumexp(s, -1)
    err = (out.float().cpu().double() - ref).abs().max().item()
    # P is rounded to bf16 inside the kernel (as flash-attn does): abs error ~ 2^-8 * |v| scale
    assert err <= 2.5e-2, f"attention max err {err}"
    rel = ((out.float().cpu().double() - ref).norm() / ref.norm()).item()
    assert rel <= 6e-3, f"attention relL2 {rel}"
    assert (lse.cpu().double() - ref_lse).abs().max().item() <= lse_tol
    return out


@pytest.mark.parametrize("hd", [64, 72, 128])
@pytest.mark.parametrize("Lq,Lk", [(256, 256), (300, 1000), (64, 65), (33, 700)])
def test_attention_vs_oracle(hip_lib, hd, Lq, Lk):
    # a stand-alone call: the kernels fold scale*log2(e) into Q and re-round it to bf16 (2^-9 relative on the logits) -- LSE tolerance
    # 4e-3 (measured 1.3e-3 .. 1.9e-3 at these shapes for every head_dim; round 3's compiler-scheduled head_dim-64 kernel did not re-round)
    _attn_case(hip_lib, 2, 2, hd, Lq, Lk, lse_tol=4e-3)


@pytest.mark.parametrize("hd", [72, 128])
def test_attention_rescale_branch(hip_lib, hd):
    # head_dim 72 / 128 (hand-scheduled kernels): q as the model path hands it over (scale folded into its single rounding);
    # a stand-alone call re-rounds scale*q, which moves a logit of ~34 by up to 2^-9 relative: LSE bound scaled with it
    # (its LSE carries the bf16 rounding of the dominant P, which is 2^(s - M) with a bf16-exact M instead of exactly 1)
    _attn_case(hip_lib, 1, 2, hd, 128, 900, spike=True, prescaled=True, lse_tol=4e-3)
    _attn_case(hip_lib, 1, 2, hd, 128, 900, spike=True, lse_tol=5e-2)


# ----------------------------------------------------------------------------- attention with a caller-supplied score bound
def _bounded_case(hip_lib, B, H, hd, Lq, Lk, bound_at_least=0.0, ws=False, seed=11, scale_q=1.0):
    """osk_attention_fwd_bounded_bf16 (FAST body): against f64 and against the tracked-max kernel"""
    D = H * hd
    q = (rnd("q", (B, Lq, D), seed=seed).float() * (scale_q * hd ** -0.5 * 1.4426950408889634)).to(BF)   # as the model path: prescaled
    kv = rnd("kv", (B, Lk, 2 * D), seed=seed + 1)
    k, v = kv[:, :, :D], kv[:, :, D:]
    Lp = (Lk + 63) // 64 * 64
    vt = torch.empty(B, H, hd, Lp, dtype=BF, device=DEV)
    hip_lib.v_transpose(v, vt, H, hd)

    def heads(t, L):
        return t.float().cpu().view(B, L, H, hd).permute(0, 2, 1, 3)

    qh, kh, vh = heads(q, Lq), heads(k, Lk), heads(v, Lk)
    s2 = qh.double() @ kh.double().transpose(-1, -2)                       # log2 units
    # a Cauchy-Schwarz bound like the model's (|q| |k| per head), optionally looser
    bound = max(float((qh.norm(dim=-1).amax() * kh.norm(dim=-1).amax())), bound_at_least)
    assert float(s2.abs().max()) <= bound
    w = hip_lib.attention_workspace(torch.device(DEV)) if ws else None
    out_b = torch.empty(B, Lq, D, dtype=BF, device=DEV)
    lse_b = torch.empty(B, H, Lq, dtype=torch.float32, device=DEV)
    hip_lib.attention_fwd(q, k, vt, out_b, H, hd, hd ** -0.5, lse=lse_b, q_prescaled=True, workspace=w, score_bound=bound)
    out_t = torch.empty_like(out_b)
    hip_lib.attention_fwd(q, k, vt, out_t, H, hd, hd ** -0.5, q_prescaled=True, workspace=w)
    s = s2 * 0.6931471805599453
    ref = (torch.softmax(s, -1) @ vh.double()).permute(0, 2, 1, 3).reshape(B, Lq, D)
    got = out_b.float().cpu().double()
    assert torch.isfinite(got).all()
    err, rel = (got - ref).abs().max().item(), ((got - ref).norm() / ref.norm()).item()
    assert err <= 2.5e-2 and rel <= 6e-3, (err, rel, bound)
    assert (lse_b.cpu().double() - torch.logsumexp(s, -1)).abs().max().item() <= 4e-3
    # same softmax from a different reference point: the two kernels agree to the bf16 rounding of P
    assert ((got - out_t.float().cpu().double()).norm() / ref.norm()).item() <= 6e-3
    return bound


@pytest.mark.parametrize("hd", [64, 72, 128])
@pytest.mark.parametrize("Lq,Lk", [(256, 64), (300, 128), (64, 192), (257, 1024), (33, 4096)])
def test_attention_bounded_fast_body_vs_f64(hip_lib, hd, Lq, Lk):
    _bounded_case(hip_lib, 2, 2, hd, Lq, Lk)


@pytest.mark.parametrize("hd", [72, 128])
def test_attention_bounded_loose_bound_tail_split_and_fallbacks(hip_lib, hd):
    b = _bounded_case(hip_lib, 1, 3, hd, 512, 2048, bound_at_least=52.0, ws=True)   # loose bound (every P ~ 2^-50), tail split + merge
    assert b == 52.0
    _bounded_case(hip_lib, 2, 2, hd, 200, 1000)                                   # ragged last tile (round 4: the FAST body's event code)
    _bounded_case(hip_lib, 1, 2, hd, 128, 512, bound_at_least=300.0)              # bound > 56: the general body runs
    _bounded_case(hip_lib, 1, 2, hd, 128, 512, scale_q=2.5)                       # larger logits, tight bound


def _bounded_segments_case(hip_lib, B, H, hd, Lq, seg, nseg, ws=False, Bkv=None, seed=61, expect_fast=True):
    """osk_attention_fwd_bounded_bf16 over `nseg` key segments of `seg` keys (any length: ragged segment-last tiles): the FAST
    body's loader events (tools/gen_attn_asm.py::fast_events) -- clamped K offsets, validity-mask ones row, segment jumps --
    against f64 and against the tracked-max (general) body on the same operands."""
    D, nb = H * hd, Bkv or B
    q = (rnd("q", (B, Lq, D), seed=seed).float() * (hd ** -0.5 * 1.4426950408889634)).to(BF)
    k = rnd("k", (nseg, nb, seg, D), seed=seed + 1)
    v = rnd("v", (nseg, nb, seg, D), seed=seed + 2)
    segp = (seg + 63) // 64 * 64
    vts = torch.empty(nseg, nb, H, hd, segp, dtype=BF, device=DEV)
    hip_lib.v_transpose(v.view(nseg * nb, seg, D), vts.view(nseg * nb, H, hd, segp), H, hd)
    kk = k.permute(1, 0, 2, 3).reshape(nb, nseg * seg, H, hd).permute(0, 2, 1, 3).double()     # [nb, H, L, hd]
    vv = v.permute(1, 0, 2, 3).reshape(nb, nseg * seg, H, hd).permute(0, 2, 1, 3).double()
    qh = q.view(B, Lq, H, hd).permute(0, 2, 1, 3).double()
    bound = float(qh.norm(dim=-1).amax() * kk.norm(dim=-1).amax())
    body = hip_lib.attention_body(hd, nseg, seg, bound)
    assert ("FAST" in body) == expect_fast, body
    args = dict(n_seg=nseg, seg_len=seg, k_seg_stride=k.stride(0), vt_seg_stride=vts.stride(0), kv_batches=Bkv or 0, q_prescaled=True,
                workspace=hip_lib.attention_workspace(torch.device(DEV)) if ws else None)
    out_b, out_t = torch.empty(B, Lq, D, dtype=BF, device=DEV), torch.empty(B, Lq, D, dtype=BF, device=DEV)
    lse_b = torch.empty(B, H, Lq, dtype=torch.float32, device=DEV)
    hip_lib.attention_fwd(q, k[0], vts, out_b, H, hd, hd ** -0.5, lse=lse_b, score_bound=bound, **args)
    hip_lib.attention_fwd(q, k[0], vts, out_t, H, hd, hd ** -0.5, **args)
    idx = torch.arange(B, device=DEV) % nb
    s = (qh @ kk[idx].transpose(-1, -2)) * 0.6931471805599453
    ref = (torch.softmax(s, -1) @ vv[idx]).permute(0, 2, 1, 3).reshape(B, Lq, D)
    got = out_b.double()
    assert torch.isfinite(got).all()
    err, rel = (got - ref).abs().max().item(), ((got - ref).norm() / ref.norm()).item()
    assert err <= 2.5e-2 and rel <= 6e-3, (err, rel, body)
    assert (lse_b.double() - torch.logsumexp(s, -1)).abs().max().item() <= 4e-3
    assert ((got - out_t.double()).norm() / ref.norm()).item() <= 6e-3


@pytest.mark.parametrize("hd", [64, 72, 128])
@pytest.mark.parametrize("seg,nseg", [(1000, 1), (60, 1), (65, 1), (130, 1), (190, 1), (4133, 1), (192, 3), (200, 2), (260, 4), (320, 2), (129, 5)])
def test_attention_bounded_ragged_tiles_and_segments_run_the_fast_body(hip_lib, hd, seg, nseg):
    """VERDICT r3 'missing' 3: ragged key counts (one tile, the prologue's tiles, both loop bodies) and several key segments
    (whole-tile and ragged, 3 .. 5 tiles each, odd and even tile counts so that events fall into both bodies)"""
    _bounded_segments_case(hip_lib, 2, 2, hd, 300, seg, nseg)


@pytest.mark.parametrize("hd", [72, 128])
def test_attention_bounded_short_segments_fall_back_and_shared_batches_tail_split(hip_lib, hd):
    _bounded_segments_case(hip_lib, 2, 2, hd, 130, 100, 3, expect_fast=False)          # several segments of 2 tiles: the general body
    _bounded_segments_case(hip_lib, 2, 2, hd, 130, 40, 2, expect_fast=False)           # ... of 1 tile
    _bounded_segments_case(hip_lib, 6, 3, hd, 4000, 300, 2, ws=True, Bkv=2)            # head-exchange call shape + tail split by segments
    _bounded_segments_case(hip_lib, 2, 9, hd, 4000, 203, 4, ws=True)                   # ragged segments as tail parts
    _bounded_segments_case(hip_lib, 1, 17, hd, 4096, 1000, 1, ws=True)                 # tail parts = tile runs, the last one ragged


@pytest.mark.parametrize("Lq,seg,nseg", [(1024, 64, 1), (1024, 128, 1), (1100, 192, 1), (1536, 1000, 1), (2000, 4133, 1), (1024, 60, 1),
                                         (1300, 130, 1), (1024, 192, 3), (2048, 200, 2), (1111, 260, 4), (1024, 129, 5)])
@pytest.mark.parametrize("hd", [72, 64])
def test_attention_wide_layout_head_dim_72(hip_lib, Lq, seg, nseg, hd):
    """attn_asm72w_kernel (bounded calls with Lq >= 1024: 512-row workgroups, one 32-key half per loop body): 1, 2, 3 and many
    key tiles (prologue only / each of the four bodies as the last one), ragged query blocks, ragged keys, segments; head_dim 64
    runs the same kernel with zero dims 64..71"""
    _bounded_segments_case(hip_lib, 2, 3, hd, Lq, seg, nseg, seed=71)


def test_attention_wide_layout_tail_split_and_determinism(hip_lib):
    _bounded_segments_case(hip_lib, 1, 16, 72, 8448, 1024, 1, ws=True, seed=73)         # 272 units: 16 tail units split by tile runs
    _bounded_segments_case(hip_lib, 3, 16, 72, 2048, 700, 3, ws=True, seed=74)          # segments as tail parts
    hd, B, H, Lq, Lk = 72, 2, 8, 2000, 4133
    D = H * hd
    q = (rnd("q", (B, Lq, D), seed=81).float() * (hd ** -0.5 * 1.4426950408889634)).to(BF)
    k, v = rnd("k", (B, Lk, D), seed=82), rnd("v", (B, Lk, D), seed=83)
    vt = torch.empty(B, H, hd, (Lk + 63) // 64 * 64, dtype=BF, device=DEV)
    hip_lib.v_transpose(v, vt, H, hd)
    outs = []
    for _ in range(12):
        o = torch.empty(B, Lq, D, dtype=BF, device=DEV)
        hip_lib.attention_fwd(q, k, vt, o, H, hd, hd ** -0.5, q_prescaled=True, score_bound=40.0)
        outs.append(o)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:]), "wide attention: run-to-run difference"


@pytest.mark.parametrize("hd,H,L", [(72, 16, 8828), (128, 6, 8828)])
def test_attention_bounded_reference_256px_length(hip_lib, hd, H, L):
    """the reference's own 256 px shape (configs/diffusion/inference/256px.py: L = 8,316 + 512 = 137 x 64 + 60)"""
    _bounded_segments_case(hip_lib, 1, H, hd, 1100, L, 1)


# the hand-scheduled kernels (attention_asm72.hip, attention_asm128.hip) at whole 64-key tiles: 1, 2, 3 and many key
# tiles (prologue-only, one body, both bodies of the 2x unrolled loop), ragged query blocks, several heads/batches
@pytest.mark.parametrize("hd", [72, 128])
@pytest.mark.parametrize("Lq,Lk", [(256, 64), (100, 128), (300, 192), (512, 1024), (33, 704), (257, 4096)])
def test_attention_asm_whole_tiles(hip_lib, Lq, Lk, hd):
    # the kernel folds scale*log2(e) into Q and re-rounds it to bf16 (2^-9 relative on the logits): LSE tolerance
    # 4e-3 instead of 2e-3; the output bounds are the common ones
    _attn_case(hip_lib, 2, 3, hd, Lq, Lk, seed=11, lse_tol=4e-3)
    # the model path: scale*log2(e) folded into q upstream -> no extra rounding, the common LSE bound holds
    _attn_case(hip_lib, 2, 3, hd, Lq, Lk, seed=12, prescaled=True, lse_tol=4e-3)


@pytest.mark.parametrize("hd", [64])
def test_attention_prescaled_q_other_head_dims(hip_lib, hd):
    _attn_case(hip_lib, 1, 2, hd, 200, 333, seed=5, prescaled=True)


@pytest.mark.parametrize("hd", [64, 72, 128])
@pytest.mark.parametrize("spike_key", [5, 64 + 7, 128 + 63, 448 + 1, 959])
def test_attention_asm_reference_max_jump(hip_lib, spike_key, hd):
    """one key whose score exceeds every earlier one by far more than the kernel's 2^8 deferral threshold: the
    reference-max move (rescale O, shift pending scores, rewrite the padding dim) must fire in the prologue
    (tile 0), in the even and in the odd loop body and in the last tile, and leave the result unchanged."""
    B, H, Lq, Lk = 1, 2, 192, 960
    D = H * hd
    q = rnd("q", (B, Lq, D), seed=21)
    kv = rnd("kv", (B, Lk, 2 * D), seed=22)
    kv[:, spike_key, :D] = q[:, 3, :] * 6.0       # row 3 (and its head neighbours) see a huge score at spike_key
    kv[:, (spike_key + 300) % Lk, :D] = q[:, 130, :] * 9.0
    k, v = kv[:, :, :D], kv[:, :, D:]
    vt = torch.empty(B, H, hd, Lk, dtype=BF, device=DEV)
    hip_lib.v_transpose(v, vt, H, hd)
    out = torch.empty(B, Lq, D, dtype=BF, device=DEV)
    lse = torch.empty(B, H, Lq, dtype=torch.float32, device=DEV)
    hip_lib.attention_fwd(q, k, vt, out, H, hd, hd ** -0.5, lse=lse)
    qh = q.float().cpu().view(B, Lq, H, hd).permute(0, 2, 1, 3).double()
    kh = k.float().cpu().view(B, Lk, H, hd).permute(0, 2, 1, 3).double()
    vh = v.float().cpu().view(B, Lk, H, hd).permute(0, 2, 1, 3).double()
    s = (qh @ kh.transpose(-1, -2)) * hd ** -0.5
    ref = (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3).reshape(B, Lq, D)
    assert (s.max(-1).values.max() > 40)          # the spike really is far above the 2^8 threshold
    # logits of ~57 (head_dim 72) / ~100 (128) carry the 2^-9 relative error of the re-rounded, pre-scaled Q of a
    # stand-alone call: bounds scale with max |s|
    assert (out.float().cpu().double() - ref).abs().max().item() <= max(3e-2, 4.5e-4 * s.abs().max().item())
    assert (lse.cpu().double() - torch.logsumexp(s, -1)).abs().max().item() <= 1.5e-3 * s.abs().max().item()


@pytest.mark.parametrize("hd", [64, 72, 128])
@pytest.mark.parametrize("seg,nseg", [(100, 3), (40, 2), (200, 2)])
def test_attention_asm_ragged_segments(hip_lib, seg, nseg, hd):
    """hand-scheduled kernel with ragged key segments (sequence-parallel all-gather layout with L/P % 64 != 0): every
    segment's last tile re-fetches the segment's last key for the missing rows and masks them out of the denominator."""
    B, H = 2, 2
    D, L = H * hd, seg * nseg
    q = rnd("q", (B, 130, D), seed=31)
    kv = rnd("kv", (B, L, 2 * D), seed=32)
    k, v = kv[:, :, :D], kv[:, :, D:]
    segp = (seg + 63) // 64 * 64
    kseg = torch.stack([k[:, i * seg:(i + 1) * seg].contiguous() for i in range(nseg)])
    vts = torch.empty(nseg, B, H, hd, segp, dtype=BF, device=DEV)
    for s_ in range(nseg):
        hip_lib.v_transpose(v[:, s_ * seg: (s_ + 1) * seg], vts[s_], H, hd)
    out = torch.empty(B, 130, D, dtype=BF, device=DEV)
    lse = torch.empty(B, H, 130, dtype=torch.float32, device=DEV)
    hip_lib.attention_fwd(q, kseg[0], vts, out, H, hd, hd ** -0.5, lse=lse, n_seg=nseg, seg_len=seg,
                          k_seg_stride=kseg.stride(0), vt_seg_stride=vts.stride(0))
    qh = q.float().cpu().view(B, 130, H, hd).permute(0, 2, 1, 3).double()
    kh = k.float().cpu().view(B, L, H, hd).permute(0, 2, 1, 3).double()
    vh = v.float().cpu().view(B, L, H, hd).permute(0, 2, 1, 3).double()
    s = (qh @ kh.transpose(-1, -2)) * hd ** -0.5
    ref = (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3).reshape(B, 130, D)
    assert (out.float().cpu().double() - ref).abs().max().item() <= 2.5e-2
    assert (lse.cpu().double() - torch.logsumexp(s, -1)).abs().max().item() <= 4e-3


@pytest.mark.parametrize("hd", [72, 128])
def test_attention_shared_key_batches_and_segments(hip_lib, hd):
    """the head-parallel sequence-parallel call shape (seqpar.py): P * B query "batches" of L/P rows each, sharing B
    key/value sets that arrive as P segments (kv_batches = B) == ordinary attention of the re-assembled sequences."""
    P, B, H, Lloc = 3, 2, 2, 128
    D, L = H * hd, P * Lloc
    q = rnd("q", (B, L, D), seed=41)
    kv = rnd("kv", (B, L, 2 * D), seed=42)
    k, v = kv[:, :, :D], kv[:, :, D:]
    vt = torch.empty(B, H, hd, L, dtype=BF, device=DEV)
    hip_lib.v_transpose(v, vt, H, hd)
    ref = torch.empty(B, L, D, dtype=BF, device=DEV)
    hip_lib.attention_fwd(q, k, vt, ref, H, hd, hd ** -0.5)
    # received-chunk layout [P, B, Lloc, D]
    to_chunks = lambda x: x.view(B, P, Lloc, D).permute(1, 0, 2, 3).contiguous()
    qr, kr, vr = to_chunks(q), to_chunks(k), to_chunks(v)
    vts = torch.empty(P, B, H, hd, Lloc, dtype=BF, device=DEV)
    hip_lib.v_transpose(vr.view(P * B, Lloc, D), vts.view(P * B, H, hd, Lloc), H, hd)
    out = torch.empty(P, B, Lloc, D, dtype=BF, device=DEV)
    hip_lib.attention_fwd(qr.view(P * B, Lloc, D), kr[0], vts, out.view(P * B, Lloc, D), H, hd, hd ** -0.5, n_seg=P,
                          seg_len=Lloc, k_seg_stride=kr.stride(0), vt_seg_stride=vts.stride(0), kv_batches=B)
    got = out.permute(1, 0, 2, 3).reshape(B, L, D)
    assert torch.equal(got, ref)   # same tiles in the same order for every query row: bit-identical


@pytest.mark.parametrize("hd", [72, 128])
def test_attention_asm_segments_and_in_place(hip_lib, hd):
    """hand-scheduled kernels: keys split in 3 segments of 128 (sequence-parallel all-gather layout) == one segment;
    output may overwrite the dead v slot."""
    B, H, L = 2, 2, 384
    D = H * hd
    y = rnd("y72", (B, L, 3 * D))
    q, k, v = y[:, :, :D], y[:, :, D: 2 * D], y[:, :, 2 * D:]
    vt = torch.empty(B, H, hd, L, dtype=BF, device=DEV)
    hip_lib.v_transpose(v, vt, H, hd)
    ref = torch.empty(B, L, D, dtype=BF, device=DEV)
    hip_lib.attention_fwd(q, k, vt, ref, H, hd, hd ** -0.5)
    seg = 128
    kseg = torch.stack([k[:, i * seg:(i + 1) * seg].contiguous() for i in range(3)])
    vts = torch.empty(3, B, H, hd, seg, dtype=BF, device=DEV)
    for s_ in range(3):
        hip_lib.v_transpose(v[:, s_ * seg: (s_ + 1) * seg], vts[s_], H, hd)
    out2 = torch.empty(B, L, D, dtype=BF, device=DEV)
    hip_lib.attention_fwd(q, kseg[0], vts, out2, H, hd, hd ** -0.5, n_seg=3, seg_len=seg,
                          k_seg_stride=kseg.stride(0), vt_seg_stride=vts.stride(0))
    assert torch.equal(out2, ref)                 # same tiles in the same order: bit-identical
    hip_lib.attention_fwd(q, k, vt, v, H, hd, hd ** -0.5)
    assert torch.equal(v, ref)


def test_attention_in_place_v_slot_and_segments(hip_lib):
    """(1) output may overwrite the dead v slot of the projection buffer; (2) keys split in 2 segments
    (sequence-parallel all-gather layout) give the same result as one segment."""
    B, H, hd, L = 1, 2, 128, 384
    D = H * hd
    y = rnd("y", (B, L, 3 * D))
    q, k, v = y[:, :, :D], y[:, :, D: 2 * D], y[:, :, 2 * D:]
    vt = torch.empty(B, H, hd, L, dtype=BF, device=DEV)
    hip_lib.v_transpose(v, vt, H, hd)
    ref = torch.empty(B, L, D, dtype=BF, device=DEV)
    hip_lib.attention_fwd(q, k, vt, ref, H, hd, hd ** -0.5)
    # two segments of 192 keys: K segments contiguous copies, VT per segment (192 -> padded 192)
    seg = 192
    kseg = torch.stack([k[:, :seg].contiguous(), k[:, seg:].contiguous()])          # [2, B, seg, D]
    vts = torch.empty(2, B, H, hd, seg, dtype=BF, device=DEV)
    for s in range(2):
        hip_lib.v_transpose(v[:, s * seg: (s + 1) * seg], vts[s], H, hd)
    out2 = torch.empty(B, L, D, dtype=BF, device=DEV)
    hip_lib.attention_fwd(q, kseg[0], vts, out2, H, hd, hd ** -0.5, n_seg=2, seg_len=seg,
                          k_seg_stride=kseg.stride(0), vt_seg_stride=vts.stride(0))
    assert (out2.float() - ref.float()).abs().max().item() <= 8e-3
    hip_lib.attention_fwd(q, k, vt, v, H, hd, hd ** -0.5)  # in place into the v slot
    assert torch.equal(v, ref)


def test_attention_constant_v_property(hip_lib):
    """softmax rows sum to one: V == c per channel  =>  out == c exactly (up to bf16), at a large shape."""
    B, H, hd, L = 1, 4, 72, 4096 + 37
    D = H * hd
    q, k = rnd("q", (B, L, D)), rnd("k", (B, L, D))
    c = rnd("c", (D,))
    v = c[None, None].expand(B, L, D).contiguous()
    Lp = (L + 63) // 64 * 64
    vt = torch.empty(B, H, hd, Lp, dtype=BF, device=DEV)
    hip_lib.v_transpose(v, vt, H, hd)
    out = torch.empty(B, L, D, dtype=BF, device=DEV)
    hip_lib.attention_fwd(q, k, vt, out, H, hd, hd ** -0.5)
    assert (out.float() - c.float()[None, None]).abs().max().item() <= 2 ** -6 * c.float().abs().max().item() + 1e-3


@pytest.mark.parametrize("hd,H", [(72, 16), (128, 6)])
def test_attention_sequence_parallel_rank_shape_720p(hip_lib, hd, H):
    """BASELINE configs[3] per-rank call shape (51 x 720p latent, SP = 8): 23,014 query rows of one rank against the
    184,112 gathered keys, 8 key segments of 23,014 (ragged last tile in every segment).  Full size on the key axis --
    this is where a 32-bit offset would wrap (K / V^T are 0.4 GB each) -- a slice of the query rows; checked against
    fp64 on sample rows and through the constant-V property on all of them."""
    P, Lloc, Lq = 8, 23014, 1100
    L, D = P * Lloc, H * hd
    g = torch.Generator(device=DEV).manual_seed(77)
    q = (torch.randn(1, Lq, D, device=DEV, generator=g)).to(BF)
    k = (torch.randn(P, 1, Lloc, D, device=DEV, generator=g)).to(BF)
    v = (torch.randn(P, 1, Lloc, D, device=DEV, generator=g)).to(BF)
    segp = (Lloc + 63) // 64 * 64
    vts = torch.empty(P, 1, H, hd, segp, dtype=BF, device=DEV)
    hip_lib.v_transpose(v.view(P, Lloc, D), vts.view(P, H, hd, segp), H, hd)
    out = torch.empty(1, Lq, D, dtype=BF, device=DEV)
    lse = torch.empty(1, H, Lq, dtype=torch.float32, device=DEV)
    hip_lib.attention_fwd(q, k[0], vts, out, H, hd, hd ** -0.5, lse=lse, n_seg=P, seg_len=Lloc,
                          k_seg_stride=k.stride(0), vt_seg_stride=vts.stride(0))
    rows = torch.tensor([0, 31, 255, 256, 1023, 1099], device=DEV)
    qh = q[0, rows].double().view(len(rows), H, hd).permute(1, 0, 2)                      # [H, r, hd]
    kh = k.view(L, H, hd).double().permute(1, 2, 0)                                        # [H, hd, L]
    s_ = (qh @ kh) * hd ** -0.5
    ref = (torch.softmax(s_, -1) @ v.view(L, H, hd).double().permute(1, 0, 2)).permute(1, 0, 2).reshape(len(rows), D)
    assert (out[0, rows].double() - ref).abs().max().item() <= 2.5e-2
    assert (lse[0][:, rows].double() - torch.logsumexp(s_, -1)).abs().max().item() <= 4e-3
    # constant V: softmax rows sum to one over all 184,112 keys, for every query row
    c = (torch.randn(D, device=DEV, generator=g)).to(BF)
    hip_lib.v_transpose(c[None, None].expand(P, Lloc, D).contiguous(), vts.view(P, H, hd, segp), H, hd)
    hip_lib.attention_fwd(q, k[0], vts, out, H, hd, hd ** -0.5, n_seg=P, seg_len=Lloc,
                          k_seg_stride=k.stride(0), vt_seg_stride=vts.stride(0))
    assert (out.float() - c.float()[None, None]).abs().max().item() <= 2 ** -6 * c.float().abs().max().item() + 1e-3


# ----------------------------------------------------------------------------- device-derived score bound (round 6)
@pytest.mark.parametrize("hd,H", [(72, 8), (64, 6), (128, 4)])
def test_rownorm2_max(hip_lib, hd, H):
    B, L = 3, 1000
    D = H * hd
    buf = rnd("x", (B, L + 3, D + 64), std=1.3, seed=301)
    x = buf[:, 3:, :D]                                            # a strided view
    out = torch.full((B, H), -1.0, dtype=torch.float32, device=DEV)
    hip_lib.rownorm2_max(x, out, H, hd)
    ref = x.float().reshape(B, L, H, hd).pow(2).sum(-1).amax(1)
    assert (out - ref).abs().max().item() <= 1e-4 * ref.max().item()
    out2 = out.clone() * 0 + 1e9                                   # accumulate keeps what is there
    hip_lib.rownorm2_max(x, out2, H, hd, accumulate=True)
    assert (out2 == 1e9).all()


def _auto_case(hip_lib, hd, H, B, Lq, Lk, scales, seed, n_seg=1, kv_batches=0):
    """q, k as the model makes them (unit RMS x per-(batch, head) scale, q pre-scaled); per (batch, head) `scales[b][h]` sets the size
    of |q| |k| -- some pairs below the FAST limit, some above -- and the auto-dispatched pair must equal fp64 on every one of them"""
    D = H * hd
    g = torch.Generator(device=DEV).manual_seed(seed)
    Bkv = kv_batches or B
    seg = Lk // n_seg

    def unit(t):
        sh = t.shape
        t = t.view(*sh[:-1], H, hd)
        return (t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-6)).view(sh)

    sc = torch.tensor(scales, dtype=torch.float32, device=DEV)                       # [rows, H]
    sc = sc.repeat((B + sc.shape[0] - 1) // sc.shape[0], 1)[:B]                       # one row per query batch
    q = unit(torch.randn(B, Lq, D, device=DEV, generator=g)).view(B, Lq, H, hd) * sc[:, None, :, None] * (hd ** -0.5 * 1.4426950408889634)
    q = q.view(B, Lq, D).to(BF)
    k = (unit(torch.randn(n_seg, Bkv, seg, D, device=DEV, generator=g)).view(n_seg, Bkv, seg, H, hd) * sc[None, :Bkv, None, :, None]).view(n_seg, Bkv, seg, D).to(BF)
    v = torch.randn(n_seg, Bkv, seg, D, device=DEV, generator=g).to(BF)
    segp = (seg + 63) // 64 * 64
    vt = torch.zeros(n_seg, Bkv, H, hd, segp, dtype=BF, device=DEV)
    hip_lib.v_transpose(v.view(n_seg * Bkv, seg, D), vt.view(n_seg * Bkv, H, hd, segp), H, hd)
    n2 = torch.empty(2, max(B, Bkv), H, dtype=torch.float32, device=DEV)
    hip_lib.rownorm2_max(q, n2[0, :B], H, hd)
    for s_ in range(n_seg):
        hip_lib.rownorm2_max(k[s_], n2[1, :Bkv], H, hd, accumulate=s_ > 0)
    bounds = (n2[0, :B].view(B, H) * n2[1, :Bkv].view(Bkv, H).repeat(B // Bkv, 1)).sqrt()
    out = torch.empty(B, Lq, D, dtype=BF, device=DEV)
    lse = torch.empty(B, H, Lq, dtype=torch.float32, device=DEV)
    ws = hip_lib.attention_workspace(q.device)
    hip_lib.attention_fwd_auto(q, k[0], vt, out, H, hd, hd ** -0.5, n2[0, :B].contiguous(), n2[1, :Bkv].contiguous(), lse=lse, n_seg=n_seg, seg_len=seg,
                               k_seg_stride=k.stride(0), vt_seg_stride=vt.stride(0), kv_batches=kv_batches, workspace=ws)
    kk = k.permute(1, 0, 2, 3).reshape(Bkv, n_seg * seg, H, hd).double().permute(0, 2, 3, 1).repeat(B // Bkv, 1, 1, 1)      # [B, H, hd, Lk]
    vv = v.permute(1, 0, 2, 3).reshape(Bkv, n_seg * seg, H, hd).double().permute(0, 2, 1, 3).repeat(B // Bkv, 1, 1, 1)
    s2 = q.double().view(B, Lq, H, hd).permute(0, 2, 1, 3) @ kk                                                            # log2 units
    p_ = torch.softmax(s2 * 0.6931471805599453, -1)
    ref = (p_ @ vv).permute(0, 2, 1, 3).reshape(B, Lq, D)
    err = (out.double() - ref).abs().max().item()
    assert err <= 2.5e-2, err
    assert (lse.double() - torch.logsumexp(s2 * 0.6931471805599453, -1)).abs().max().item() <= 6e-3
    return bounds, out


@pytest.mark.parametrize("hd,H", [(72, 4), (64, 4), (128, 4)])
def test_attention_auto_bound_mixes_fast_and_general_units(hip_lib, hd, H):
    """osk_attention_fwd_auto_bf16: the bound comes from the operands, per (batch, head).  Scales chosen so that the (batch, head)
    pairs straddle the FAST limit 56 (bound = 1.44 sqrt(hd) s^2 up to rounding): both kernels of the launch pair work, each on its
    own units, the result equals fp64 everywhere -- wide units (Lq >= 1024), a ragged key tile, a tail split, segments, kv_batches."""
    c = hd ** 0.5 * 1.4426950408889634
    lo, hi = (20.0 / c) ** 0.5, (90.0 / c) ** 0.5                       # bounds ~20 (FAST) and ~90 (general)
    scales = [[lo, hi, lo, hi][:H], [hi, hi, lo, lo][:H]]
    bounds, _ = _auto_case(hip_lib, hd, H, 2, 1500, 2000 + 37, scales, seed=311)
    assert (bounds.min().item() < 56.0 < bounds.max().item()), bounds
    _auto_case(hip_lib, hd, H, 2, 300, 640, scales, seed=312)                                   # 256-row units only
    _auto_case(hip_lib, hd, H, 2, 1100, 4 * 256, scales, seed=313, n_seg=4)                     # key segments of 4 tiles
    _auto_case(hip_lib, hd, H, 4, 700, 900, scales, seed=314, kv_batches=2)                     # query batches sharing key sets
    # every pair FAST / every pair general: the twin launches nothing but early exits
    _auto_case(hip_lib, hd, H, 2, 1200, 1300, [[lo] * H, [lo] * H], seed=315)
    _auto_case(hip_lib, hd, H, 2, 1200, 1300, [[hi] * H, [hi] * H], seed=316)


def test_attention_auto_bound_adversarial_rows(hip_lib):
    """rows the weight-derived promise could not cover: one dominant key (its score sits AT the Cauchy-Schwarz bound), all keys equal
    (softmax = uniform: the output is the mean of V), a zero query row"""
    hd, H, B, Lq, Lk = 72, 4, 1, 1100, 1500
    D = H * hd
    g = torch.Generator(device=DEV).manual_seed(321)
    q = (torch.randn(B, Lq, D, device=DEV, generator=g) * 0.35).to(BF)
    k = (torch.randn(B, Lk, D, device=DEV, generator=g) * 0.9).to(BF)
    v = torch.randn(B, Lk, D, device=DEV, generator=g).to(BF)
    q[0, 7] = 0                                                   # zero query row: uniform softmax
    q[0, 9] = (k[0, 123].float() * 0.4).to(BF)                    # aligned with one key of large norm: that key dominates
    k[0, 123] = k[0, 123] * 3
    k[0, :, 2 * hd: 3 * hd] = k[0, 0, 2 * hd: 3 * hd]            # head 2: all keys equal
    vt = torch.zeros(B, H, hd, (Lk + 63) // 64 * 64, dtype=BF, device=DEV)
    hip_lib.v_transpose(v, vt, H, hd)
    n2 = torch.empty(2, B, H, dtype=torch.float32, device=DEV)
    hip_lib.rownorm2_max(q, n2[0], H, hd)
    hip_lib.rownorm2_max(k, n2[1], H, hd)
    out = torch.empty(B, Lq, D, dtype=BF, device=DEV)
    hip_lib.attention_fwd_auto(q, k, vt, out, H, hd, 1.0 / 1.4426950408889634, n2[0], n2[1], q_prescaled=True, workspace=hip_lib.attention_workspace(q.device))
    s2 = q.double().view(B, Lq, H, hd).permute(0, 2, 1, 3) @ k.double().view(B, Lk, H, hd).permute(0, 2, 3, 1)
    ref = (torch.softmax(s2 * 0.6931471805599453, -1) @ v.double().view(B, Lk, H, hd).permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(B, Lq, D)
    assert torch.isfinite(out.float()).all()
    assert (out.double() - ref).abs().max().item() <= 2.5e-2
    mean_v = v.double().view(B, Lk, H, hd).mean(1)                # head 2 (equal keys) and row 7 (zero query): the mean of V
    assert (out[0, :, 2 * hd: 3 * hd].double() - mean_v[0, 2]).abs().max().item() <= 1e-2
    assert (out[0, 7].double() - mean_v[0].reshape(D)).abs().max().item() <= 1e-2


# ----------------------------------------------------------------------------- osk_gemm_group_bf16 (round 6)
def _vt_reference(hip_lib, x, w, bias, H, hd):
    """the two-kernel path the V^T task replaces: osk_gemm_bf16 into [B, L, H*hd] + osk_v_transpose_bf16"""
    B, L, K = x.shape
    v = torch.empty(B, L, H * hd, dtype=BF, device=DEV)
    hip_lib.gemm(x, w, bias, v)
    vt = torch.zeros(B, H, hd, (L + 63) // 64 * 64, dtype=BF, device=DEV)
    hip_lib.v_transpose(v, vt, H, hd)
    return vt


@pytest.mark.parametrize("hd,H,B,L,K,with_bias", [(72, 16, 3, 1000, 1152, True), (72, 16, 1, 2048, 1152, False), (64, 6, 2, 777, 384, True),
                                                    (128, 24, 2, 1300, 3072, True), (128, 4, 1, 4096 + 64, 512, True), (72, 4, 2, 100, 320, True)])
def test_gemm_group_vt_task_equals_gemm_plus_v_transpose(hip_lib, hd, H, B, L, K, with_bias):
    """V^T task of osk_gemm_group_bf16: the V projection written directly in the attention kernels' key-major operand layout
    (per 64-key group in the order of osk_v_transpose_bf16), ragged key count (zero pad), row-batch straddling tiles, against the
    two-kernel path (one bf16 step: the bias is added after the K loop instead of before it) and against fp64."""
    N = H * hd
    x = rnd("x", (B, L, K), seed=201)
    w = rnd("w", (N, K), std=K ** -0.5, seed=202)
    bias = rnd("b", (N,), std=0.3, dtype=torch.float32, seed=203) if with_bias else None
    Lp = (L + 63) // 64 * 64
    vt = torch.full((B, H, hd, Lp + 64), 7.0, dtype=BF, device=DEV)          # a wider key axis: the task starts at position 64
    ok = hip_lib.gemm_group([dict(x=x, w=w, bias=bias, vt=vt, vt_pos=64, hd=hd)])
    assert ok
    ref = _vt_reference(hip_lib, x, w, bias, H, hd)
    assert (vt[..., :64] == 7.0).all(), "wrote in front of its first position"
    got = vt[..., 64:]
    bf16_ulp_close(got.float().cpu(), ref.float().cpu(), rel=2 ** -7, abs_=2e-3)
    # fp64: v[b, key, n] at position p of its 64-key group
    from tests import cpu_ops
    key = cpu_ops.pos2key(hd, Lp)
    v64 = x.double().cpu() @ w.double().cpu().T + (bias.double().cpu() if with_bias else 0.0)
    pad = torch.zeros(B, Lp, N, dtype=torch.float64)
    pad[:, :L] = v64
    ref64 = pad[:, key].reshape(B, Lp, H, hd).permute(0, 2, 3, 1)
    bf16_ulp_close(got.float().cpu(), ref64.float().bfloat16().float(), rel=2 ** -7, abs_=2e-3)
    valid = (key < L)
    assert (got[..., ~valid.to(DEV)] == 0).all(), "positions behind the sequence end must be zero"


def test_gemm_group_skip_range_and_block_packs(hip_lib):
    """a single-stream block's linear1 without its V columns (row layout [q | k | . | gelu(mlp)]) + the V^T task in one launch, and a
    double block's four problems (img / txt q|k + img / txt V^T behind each other on the key axis) -- against the single calls."""
    H, hd, B, Lt, Li = 16, 72, 2, 128, 1100
    D, R, L = H * hd, 4 * H * hd, 128 + 1100
    xm = rnd("xm", (B, L, D), seed=211)
    w1 = rnd("w1", (3 * D + R, D), std=D ** -0.5, seed=212)
    b1 = rnd("b1", (3 * D + R,), std=0.2, dtype=torch.float32, seed=213)
    y_ref = torch.empty(B, L, 3 * D + R, dtype=BF, device=DEV)
    hip_lib.gemm(xm, w1, b1, y_ref, gelu_from=3 * D)
    vt_ref = _vt_reference(hip_lib, xm, w1[2 * D: 3 * D], b1[2 * D: 3 * D], H, hd)
    y = torch.full_like(y_ref, 3.0)
    Lp = (L + 63) // 64 * 64
    vt = torch.empty(B, H, hd, Lp, dtype=BF, device=DEV)
    assert hip_lib.gemm_group([dict(a=xm, w=w1, bias=b1, out=y, gelu_from=3 * D, skip=(2 * D, D)),
                               dict(x=xm, w=w1[2 * D: 3 * D], bias=b1[2 * D: 3 * D], vt=vt, vt_pos=0, hd=hd)])
    assert (y[:, :, 2 * D: 3 * D] == 3.0).all(), "the skipped columns were written"
    assert torch.equal(y[:, :, :2 * D], y_ref[:, :, :2 * D]) and torch.equal(y[:, :, 3 * D:], y_ref[:, :, 3 * D:])
    bf16_ulp_close(vt.float().cpu(), vt_ref.float().cpu(), rel=2 ** -7, abs_=2e-3)
    # double block: joint [txt ; img] rows, per-stream weights
    wq = {s_: rnd("wqkv" + s_, (3 * D, D), std=D ** -0.5, seed=214 + i) for i, s_ in enumerate(("img", "txt"))}
    bq = {s_: rnd("bqkv" + s_, (3 * D,), std=0.2, dtype=torch.float32, seed=216 + i) for i, s_ in enumerate(("img", "txt"))}
    y3_ref = torch.empty(B, L, 3 * D, dtype=BF, device=DEV)
    hip_lib.gemm(xm[:, Lt:], wq["img"], bq["img"], y3_ref[:, Lt:])
    hip_lib.gemm(xm[:, :Lt], wq["txt"], bq["txt"], y3_ref[:, :Lt])
    vt2_ref = torch.zeros(B, H, hd, Lp, dtype=BF, device=DEV)
    hip_lib.v_transpose(y3_ref[:, :, 2 * D:], vt2_ref, H, hd)
    y3 = torch.full_like(y3_ref, 5.0)
    vt2 = torch.empty(B, H, hd, Lp, dtype=BF, device=DEV)
    tasks = []
    for s_, rows, pos in (("img", slice(Lt, L), Lt), ("txt", slice(0, Lt), 0)):
        tasks.append(dict(a=xm[:, rows], w=wq[s_][:2 * D], bias=bq[s_][:2 * D], out=y3[:, rows]))
        tasks.append(dict(x=xm[:, rows], w=wq[s_][2 * D:], bias=bq[s_][2 * D:], vt=vt2, vt_pos=pos, hd=hd))
    assert hip_lib.gemm_group(tasks)
    assert (y3[:, :, 2 * D:] == 5.0).all()
    assert torch.equal(y3[:, :, :2 * D], y3_ref[:, :, :2 * D])
    bf16_ulp_close(vt2.float().cpu(), vt2_ref.float().cpu(), rel=2 ** -7, abs_=2e-3)
    # repeatable
    vt3 = torch.empty_like(vt2)
    for t_ in tasks:
        if "vt" in t_:
            t_["vt"] = vt3
    assert hip_lib.gemm_group(tasks) and torch.equal(vt3, vt2)


def test_gemm_group_declines_shapes_off_the_large_tile_path(hip_lib):
    """OSK_EUNSUPPORTED -> False, NOTHING launched (the caller then runs the single calls); bad skip ranges are errors"""
    x = rnd("x", (1, 100, 128), seed=221)
    w = rnd("w", (128, 128), seed=222)
    out = torch.full((1, 100, 128), 9.0, dtype=BF, device=DEV)
    assert hip_lib.gemm_group([dict(a=x, w=w, bias=None, out=out)]) is False          # M < 256
    vt = torch.full((1, 2, 64, 128), 9.0, dtype=BF, device=DEV)
    assert hip_lib.gemm_group([dict(x=x, w=w, bias=None, vt=vt, vt_pos=0, hd=64)]) is False     # H * hd < 256
    torch.cuda.synchronize()
    assert (out == 9.0).all() and (vt == 9.0).all()
    x2 = rnd("x2", (1, 512, 128), seed=223)
    w2 = rnd("w2", (1024, 128), seed=224)
    out2 = torch.empty(1, 512, 1024, dtype=BF, device=DEV)
    with pytest.raises(RuntimeError):
        hip_lib.gemm_group([dict(a=x2, w=w2, bias=None, out=out2, skip=(128, 256))])     # skip_from % 256 != 0


# ----------------------------------------------------------------------------- race screens for the asm K loops
def _attention_determinism(hip_lib, hd):
    # ragged keys + ragged query block, several rounds of workgroups
    B, H, Lq, Lk = 2, 8, 2000, 4133
    D = H * hd
    q, k, v = rnd("q", (B, Lq, D), seed=51), rnd("k", (B, Lk, D), seed=52), rnd("v", (B, Lk, D), seed=53)
    vt = torch.empty(B, H, hd, (Lk + 63) // 64 * 64, dtype=BF, device=DEV)
    hip_lib.v_transpose(v, vt, H, hd)
    outs = []
    for _ in range(12):
        o = torch.empty(B, Lq, D, dtype=BF, device=DEV)
        hip_lib.attention_fwd(q, k, vt, o, H, hd, hd ** -0.5)
        outs.append(o)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:]), "attention: run-to-run difference"
    rows = torch.tensor([0, 1, 255, 256, 1023, 1999])
    qh = q[:, rows].float().cpu().view(B, len(rows), H, hd).permute(0, 2, 1, 3).double()
    kh = k.float().cpu().view(B, Lk, H, hd).permute(0, 2, 1, 3).double()
    vh = v.float().cpu().view(B, Lk, H, hd).permute(0, 2, 1, 3).double()
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * hd ** -0.5, -1) @ vh).permute(0, 2, 1, 3).reshape(B, len(rows), D)
    assert (outs[0][:, rows].float().cpu().double() - ref).abs().max().item() <= 2.5e-2


def test_attention_asm128_is_deterministic_under_load(hip_lib):
    torch.manual_seed(0)
    _attention_determinism(hip_lib, 128)


def test_hand_scheduled_kernels_are_deterministic_under_load(hip_lib):
    """The generated asm loops place their own s_waitcnt / barriers / hazard gaps: a missed one shows up as rare,
    load-dependent wrong tiles, not as a steady error.  Run each kernel 12 times at a shape that fills the chip for
    several rounds (uneven tail included) and require bit-identical results, then check one full result against a
    float64 reference on a sample of rows."""
    torch.manual_seed(0)
    _attention_determinism(hip_lib, 72)
    # --- large-tile GEMM (ragged M and N tiles, 18 K steps)
    a = rnd("a", (3, 2100, 1152), seed=54)
    w = rnd("w", (1160, 1152), std=1152 ** -0.5, seed=55)
    bias = rnd("b", (1160,), std=0.1, dtype=torch.float32, seed=56)
    gs = []
    for _ in range(12):
        o = torch.empty(3, 2100, 1160, dtype=BF, device=DEV)
        hip_lib.gemm(a, w, bias, o)
        gs.append(o)
    torch.cuda.synchronize()
    assert all(torch.equal(gs[0], o) for o in gs[1:]), "gemm256: run-to-run difference"
    bf16_ulp_close(gs[0].float().cpu(), _gemm_ref(a, w, bias).float().bfloat16().float(), rel=2 ** -7, abs_=2e-3)


# ----------------------------------------------------------------------------- small kernels
@pytest.mark.parametrize("Bv,K", [(3, 384), (6, 384), (2, 4104), (5, 1152)])   # x slices of 4 / 8 / 1 / 8 rows in LDS
def test_gemv_tasks_and_timestep_embedding(hip_lib, Bv, K):
    x = rnd("vec", (Bv, K), dtype=torch.float32)
    layers, col = [], 0
    refs = []
    for i, n in enumerate([6 * K, 3 * K, 100]):
        w = rnd(f"w{i}", (n, K), std=K ** -0.5)
        b = rnd(f"b{i}", (n,), std=0.1) if i != 1 else None
        layers.append((w, b, col))
        r = torch.nn.functional.silu(x.cpu().double()) @ w.cpu().double().T
        if b is not None:
            r = r + b.cpu().double()
        refs.append(r)
        col += n
    tasks = hip_lib.GemvTasks(layers, DEV)
    out = torch.zeros(Bv, col, dtype=torch.float32, device=DEV)
    hip_lib.gemv_tasks(x, tasks, out, act_in=1)
    ref = torch.cat(refs, 1)
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-4
    hip_lib.gemv_tasks(x, tasks, out, act_in=1, accumulate=True)
    assert (out.cpu().double() - 2 * ref).abs().max().item() <= 2e-4
    if (Bv, K) != (3, 384):
        return
    t = torch.tensor([0.69921875, 0.0, 1.0], dtype=torch.float32, device=DEV)
    emb = torch.empty(3, 256, dtype=torch.float32, device=DEV)
    hip_lib.timestep_embedding(t, emb)
    ref_e = O.timestep_embedding(t.cpu(), 256)
    assert (emb.cpu() - ref_e).abs().max().item() <= 2e-4  # f32 trig of arguments up to 1e3


def test_cfg_euler(hip_lib):
    n = 8 * 1000
    pred = rnd("p", (3, n))
    x = rnd("x", (n,))
    xo = torch.empty_like(x)
    hip_lib.cfg_euler(pred, x, xo, 7.5, 3.0, -0.0321)
    c, u, u2 = pred.float().cpu()
    ref = x.float().cpu() + (-0.0321) * (u2 + 3.0 * (u - u2) + 7.5 * (c - u))
    bf16_ulp_close(xo.float().cpu(), ref.bfloat16().float(), abs_=1e-3)


def test_copy_rows(hip_lib):
    """osk_copy_rows_bf16: strided row copies of the step's host glue -- a column slice of a wider operand as destination, a
    batch-broadcast source, odd row counts; bit-exact, nothing outside the destination columns touched."""
    B, L, C, Kp = 3, 1031, 68, 192
    src = rnd("s", (B, L, C))
    dst = torch.full((B, L, Kp), 7.0, dtype=BF, device=DEV)
    hip_lib.copy_rows(src, dst[:, :, 64:])
    assert torch.equal(dst[:, :, 64:64 + C], src) and bool((dst[:, :, :64] == 7).all()) and bool((dst[:, :, 64 + C:] == 7).all())
    one = rnd("o", (1, L, 64))
    out = torch.empty(B, L, 64, dtype=BF, device=DEV)
    hip_lib.copy_rows(one, out)
    assert torch.equal(out, one.expand(B, L, 64))
    wide = rnd("w", (2, 40, 256))
    view = wide[:, 3:36, 128:192]                       # strided source: row stride 256, column offset
    out2 = torch.empty(2, 33, 64, dtype=BF, device=DEV)
    hip_lib.copy_rows(view, out2)
    assert torch.equal(out2, view)
    with pytest.raises(RuntimeError):
        hip_lib.copy_rows(rnd("b", (1, 8, 6)), torch.empty(1, 8, 6, dtype=BF, device=DEV))   # C % 4 != 0


def test_errors_raise(hip_lib):
    a = torch.zeros(1, 8, 100, dtype=BF, device=DEV)
    w = torch.zeros(16, 100, dtype=BF, device=DEV)
    out = torch.zeros(1, 8, 16, dtype=BF, device=DEV)
    with pytest.raises(RuntimeError):
        hip_lib.gemm(a, w, None, out)  # K % 64 != 0


# ----------------------------------------------------------------------------- tail split (workspace variant)
@pytest.mark.parametrize("hd", [64, 72, 128])
@pytest.mark.parametrize("case", ["one_segment_ragged", "few_units", "segments", "shared_key_batches", "one_tile_per_part"])
def test_attention_tail_split_matches_unsplit(hip_lib, hd, case):
    """osk_attention_fwd_ws_bf16: the work units of the grid's last partial round are cut into key parts and merged by
    LSE.  Same inputs with and without the workspace: a split row sees its P values rounded to bf16 against the
    reference max of its own key part instead of the global one, so the two results differ at the kernel's own
    rounding level (a fraction of its 2.5e-2 bound against fp64), nowhere more; LSE within 1e-3; the split really
    happens (osk_attention_tail_split_factor > 1)."""
    kw, Bkv = {}, None
    if case == "one_segment_ragged":      # 272 units on 256 CUs: 16 tail units x 8 parts of 2 tiles (last one ragged)
        B, H, Lq, Lk, n_seg = 1, 17, 4096, 1000, 1
    elif case == "few_units":             # the whole launch is a tail: 32 units x 8 parts
        B, H, Lq, Lk, n_seg = 1, 4, 2048, 4133, 1
    elif case == "segments":              # 4 ragged key segments: parts = whole segments
        B, H, Lq, Lk, n_seg = 2, 9, 4000, 400, 4
    elif case == "shared_key_batches":    # head-parallel sequence-parallel call shape: 6 query batches share 2 key sets
        B, H, Lq, Lk, n_seg, Bkv = 6, 3, 4000, 512, 2, 2
    else:                                 # parts of a single tile each, ragged last one
        B, H, Lq, Lk, n_seg = 1, 17, 4096, 100 + 64 * 3, 1
    D = H * hd
    seg = Lk // n_seg
    q = rnd("q", (B, Lq, D), seed=101)
    nb = Bkv or B
    k = rnd("k", (n_seg, nb, seg, D), seed=102)
    v = rnd("v", (n_seg, nb, seg, D), seed=103)
    segp = (seg + 63) // 64 * 64
    vts = torch.empty(n_seg, nb, H, hd, segp, dtype=BF, device=DEV)
    hip_lib.v_transpose(v.view(n_seg * nb, seg, D), vts.view(n_seg * nb, H, hd, segp), H, hd)
    args = dict(n_seg=n_seg, seg_len=seg, k_seg_stride=k.stride(0), vt_seg_stride=vts.stride(0), kv_batches=Bkv or 0)
    ws = hip_lib.attention_workspace(q.device)
    assert hip_lib.lib.osk_attention_tail_split_factor(B, H, Lq, n_seg, seg, hd, ws.numel()) > 1
    outs, lses = [], []
    for w in (None, ws):
        o = torch.empty(B, Lq, D, dtype=BF, device=DEV)
        l = torch.empty(B, H, Lq, dtype=torch.float32, device=DEV)
        hip_lib.attention_fwd(q, k[0], vts, o, H, hd, hd ** -0.5, lse=l, workspace=w, **args)
        outs.append(o.float())
        lses.append(l)
    d = (outs[0] - outs[1]).abs()
    assert d.max().item() <= 1.2e-2, d.max().item()
    rel = (d.double().norm() / outs[0].double().norm()).item()
    assert rel <= 5e-3, rel
    assert (lses[0] - lses[1]).abs().max().item() <= 1e-3
    # and both against fp64 on sample rows of the last query block (a tail unit)
    rows = torch.tensor([Lq - 1, Lq - 77, Lq - 200], device=DEV)
    kk = k.permute(1, 0, 2, 3).reshape(nb, n_seg * seg, H, hd).double()
    vv = v.permute(1, 0, 2, 3).reshape(nb, n_seg * seg, H, hd).double()
    for b in range(B):
        qh = q[b, rows].double().view(len(rows), H, hd).permute(1, 0, 2)
        s_ = (qh @ kk[b % nb].permute(1, 2, 0)) * hd ** -0.5
        ref = (torch.softmax(s_, -1) @ vv[b % nb].permute(1, 0, 2)).permute(1, 0, 2).reshape(len(rows), D)
        for o in outs:
            assert (o[b, rows].double() - ref).abs().max().item() <= 2.5e-2
