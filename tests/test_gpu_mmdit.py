"""GPU parity of the whole denoiser: open_sora_amd.mmdit.MMDiTModel.forward (HIP kernels through the C ABI)
against (a) the committed goldens made by the REAL reference in fp32 and (b) the CPU oracle, with the tolerance
policy of SURVEY.md §8(d):  relL2(ours, fp32 truth) <= max(1.5 * relL2(reference-precision bf16, truth), 2^-8)."""
import os

import numpy as np
import pytest
import torch

from oracle import configs, mmdit_oracle as O
from tests.util import assert_parity, torch_inputs, torch_params

pytestmark = pytest.mark.gpu
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BF = torch.bfloat16


def _build(cfg, device="cuda"):
    from open_sora_amd import mmdit

    model = mmdit.Flux(device_map=device, torch_dtype=BF, **cfg)
    sd = torch_params(cfg, dtype=BF, device=device)
    model.load_state_dict(sd, strict=True)
    return model


@pytest.mark.parametrize("name", list(configs.GOLDEN))
def test_forward_matches_reference_golden(hip_lib, name):
    cfg, B, T, h, w, L_txt = configs.GOLDEN[name]
    g = np.load(os.path.join(GOLDEN_DIR, f"mmdit_{name}.npz"))
    truth = torch.from_numpy(g["out"])
    model = _build(cfg)
    inp = torch_inputs(cfg, B, T, h, w, L_txt, dtype=BF, device="cuda")
    with torch.inference_mode():
        out = model(**inp)
    assert out.shape == truth.shape and out.dtype == BF
    # reference-precision comparator: the oracle run with bf16 tensors on CPU (same rounding points as the
    # reference's bf16 eager path; pinned against it by tests/test_oracle_vs_reference.py)
    sdb = torch_params(cfg, dtype=BF)
    inpb = torch_inputs(cfg, B, T, h, w, L_txt, dtype=BF)
    with torch.inference_mode():
        ref_bf16 = O.forward(sdb, cfg, **inpb)
    assert_parity(out, truth, ref_bf16, f"MMDiT forward [{name}]")


def test_forward_is_deterministic_and_reusable(hip_lib):
    cfg, B, T, h, w, L_txt = configs.GOLDEN["hd72_eager_split"]
    model = _build(cfg)
    inp = torch_inputs(cfg, B, T, h, w, L_txt, dtype=BF, device="cuda")
    with torch.inference_mode():
        a = model(**inp).clone()
        b = model(**inp).clone()
    assert torch.equal(a, b)


@pytest.mark.parametrize("name", ["hd64_eager_fused", "hd128_liger_split"])
def test_block_processors_match_oracle(hip_lib, name):
    """The block-level plug-in: `block(img, txt, vec, pe)` with the reference's pe formats
    (EmbedND tensor [B,1,L,hd/2,2,2] / LigerEmbedND (cos, sin) tuple)."""
    cfg, B, T, h, w, L_txt = configs.GOLDEN[name]
    model = _build(cfg)
    sd32 = torch_params(cfg)
    inp32 = torch_inputs(cfg, B, T, h, w, L_txt)
    with torch.inference_mode():
        img, txt, vec, ang = O.prepare_block_inputs(sd32, cfg, **inp32)
    liger = cfg.get("use_liger_rope", False)
    if liger:
        cs = torch.cos(ang.float()).repeat(1, 1, 2).cuda()
        sn = torch.sin(ang.float()).repeat(1, 1, 2).cuda()
        pe = (cs, sn)
        mode = "half"
    else:
        c, s = torch.cos(ang), torch.sin(ang)
        pe = torch.stack([c, -s, s, c], dim=-1).reshape(*ang.shape, 2, 2).float().unsqueeze(1).cuda()
        mode = "interleaved"
    with torch.inference_mode():
        t_img, t_txt = O.double_block(sd32, cfg, 0, img, txt, vec, ang, mode)
        sdb = {k: v.bfloat16() for k, v in sd32.items()}
        r_img, r_txt = O.double_block(sdb, cfg, 0, img.bfloat16(), txt.bfloat16(), vec.bfloat16(), ang, mode)
        o_img, o_txt = model.double_blocks[0](img.bfloat16().cuda(), txt.bfloat16().cuda(), vec.bfloat16().cuda(), pe)
    assert_parity(o_img, t_img, r_img, f"double block img [{name}]")
    assert_parity(o_txt, t_txt, r_txt, f"double block txt [{name}]")
    x = torch.cat((t_txt, t_img), 1)
    with torch.inference_mode():
        t_x = O.single_block(sd32, cfg, 0, x, vec, ang, mode)
        r_x = O.single_block(sdb, cfg, 0, x.bfloat16(), vec.bfloat16(), ang, mode)
        o_x = model.single_blocks[0](x.bfloat16().cuda(), vec.bfloat16().cuda(), pe)
    assert_parity(o_x, t_x, r_x, f"single block [{name}]")


def test_error_conventions(hip_lib):
    """ValueError for ndim != 3 and for a missing cond (model.py:172-179)."""
    cfg, B, T, h, w, L_txt = configs.GOLDEN["hd64_eager_fused"]
    model = _build(cfg)
    inp = torch_inputs(cfg, B, T, h, w, L_txt, dtype=BF, device="cuda")
    bad = dict(inp)
    bad["img"] = inp["img"][0]
    with pytest.raises(ValueError):
        model(**bad)
    bad = dict(inp)
    bad.pop("cond")
    with pytest.raises(ValueError):
        model(**bad)


def test_s_config_single_step(hip_lib):
    """BASELINE config 1 geometry on the GPU: S-width MMDiT, 1x128x128 latent -> L_img 4096, L_txt 512."""
    cfg = configs.MMDIT["S"]
    model = _build(cfg)
    inp = torch_inputs(cfg, 1, 1, 64, 64, 512, dtype=BF, device="cuda")
    with torch.inference_mode():
        out = model(**inp)
    sd32 = torch_params(cfg)
    inp32 = torch_inputs(cfg, 1, 1, 64, 64, 512)
    with torch.inference_mode():
        truth = O.forward(sd32, cfg, **inp32)
        ref_bf16 = O.forward({k: v.bfloat16() for k, v in sd32.items()}, cfg,
                             **torch_inputs(cfg, 1, 1, 64, 64, 512, dtype=BF))
    assert_parity(out, truth, ref_bf16, "MMDiT-S single step (cfg 1)")


@pytest.mark.parametrize("qk_scale", [1.0, 2.5], ids=["host_bound", "device_bound"])
def test_denoise_step_is_hipgraph_capturable(hip_lib, qk_scale):
    """include/osk.h promises entry points without allocation, synchronisation or global state, i.e. capturable in a
    hipGraph: capture ONE whole denoise step (MMDiT forward on the CFG triple + the fused CFG / Euler update) with
    torch.cuda.graph, replay it on new inputs written into the captured buffers, and compare with the eager step."""
    from open_sora_amd import _C

    cfg, _, T, h, w, L_txt = configs.GOLDEN["hd72_eager_split"]
    B = 3
    model = _build(cfg)
    if qk_scale != 1.0:
        # QK-norm scales whose weight-derived bound exceeds the FAST limit: the blocks take the bound from their operands on the device
        # (osk_rownorm2_max_bf16 + the auto-dispatched launch pair, round 6) -- a memset, two reductions and two launches per block that
        # must be capturable like everything else
        with torch.no_grad():
            for blk in list(model.double_blocks) + list(model.single_blocks):
                for nrm in ([blk.img_attn.norm, blk.txt_attn.norm] if hasattr(blk, "img_attn") else [blk.norm]):
                    nrm.query_norm.scale.mul_(qk_scale)
                    nrm.key_norm.scale.mul_(qk_scale)
        model.invalidate_plan()
    inp = torch_inputs(cfg, B, T, h, w, L_txt, dtype=BF, device="cuda")
    x = inp["img"][:1].clone().contiguous()
    x_next = torch.empty_like(x)

    def step():
        inp["img"].copy_(x.expand(B, -1, -1))
        pred = model(**inp)
        _C.cfg_euler(pred, x, x_next, 7.5, 3.0, -0.05)

    with torch.inference_mode():
        step()                                   # warm-up: builds the plan and the workspaces outside the capture
        torch.cuda.synchronize()
        rep = model.attention_report()
        assert rep["blocks_auto_dispatched"] == (0 if qk_scale == 1.0 else rep["blocks"]), rep
        x0 = x.clone()
        eager = []
        for i in range(2):                       # eager results for two different latent states
            x.copy_(x0 * (1.0 + 0.5 * i))
            step()
            eager.append(x_next.clone())
        graph = torch.cuda.CUDAGraph()
        x.copy_(x0)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()                               # side-stream warm-up, as torch's capture recipe asks
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(graph):
            step()
        for i in range(2):
            x.copy_(x0 * (1.0 + 0.5 * i))
            x_next.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(x_next, eager[i]), f"hipGraph replay {i} differs from the eager step"


def test_checkpoint_load_and_rope_convention_on_gpu(hip_lib, tmp_path):
    """SURVEY section 8(f) rank 3 on the device: (1) `Flux(from_pretrained=<safetensors>)` straight onto the GPU reproduces the
    reference golden; (2) loading a NEW checkpoint into a model that has already run re-plans the kernels' weight images
    (no stale copies); (3) the eager-convention checkpoint permuted by `convert_rope_convention` and evaluated with
    `use_liger_rope=True` is the same function on the HIP path (the "fused-rope" checkpoint of docs/train.md:112)."""
    from safetensors.torch import save_file

    from open_sora_amd import ckpt, mmdit

    name = "hd72_eager_split"
    cfg, B, T, h, w, L_txt = configs.GOLDEN[name]
    sd = torch_params(cfg, dtype=BF)
    path = os.path.join(tmp_path, "model.safetensors")
    save_file({k: v.contiguous() for k, v in sd.items()}, path)
    model = mmdit.Flux(from_pretrained=path, device_map="cuda", torch_dtype=BF, strict_load=True, **cfg)
    inp = torch_inputs(cfg, B, T, h, w, L_txt, dtype=BF, device="cuda")
    truth = torch.from_numpy(np.load(os.path.join(GOLDEN_DIR, f"mmdit_{name}.npz"))["out"])
    with torch.inference_mode():
        out = model(**inp)
        ref_bf16 = O.forward(torch_params(cfg, dtype=BF), cfg, **torch_inputs(cfg, B, T, h, w, L_txt, dtype=BF))
    assert_parity(out, truth, ref_bf16, "MMDiT from a safetensors checkpoint on the GPU")
    # (2) other weights into the SAME module: the output must follow them
    sd2 = torch_params(cfg, seed=5, dtype=BF)
    p2 = os.path.join(tmp_path, "model2.safetensors")
    save_file({k: v.contiguous() for k, v in sd2.items()}, p2)
    ckpt.load_checkpoint(model, p2, strict=True)
    fresh = _build(cfg)
    fresh.load_state_dict({k: v.cuda() for k, v in sd2.items()}, strict=True)
    with torch.inference_mode():
        a, b = model(**inp), fresh(**inp)
    assert torch.equal(a, b), "weights loaded after a first forward were not picked up by the kernels' plan"
    # (3) RoPE-convention transform
    half = ckpt.convert_rope_convention(sd, cfg["hidden_size"], cfg["num_heads"], to="half")
    m_half = mmdit.Flux(device_map="cuda", torch_dtype=BF, **dict(cfg, use_liger_rope=True))
    m_half.load_state_dict({k: v.cuda() for k, v in half.items()}, strict=True)
    with torch.inference_mode():
        out_half = m_half(**inp)
    assert_parity(out_half, truth, ref_bf16, "MMDiT, permuted checkpoint + liger RoPE convention")


def test_two_models_and_two_streams_do_not_share_workspaces(hip_lib):
    """Workspaces are owned per model and per stream (round 1 kept one process-global workspace per
    geometry): two models of the same geometry driven concurrently from two streams, and ONE model driven from two
    streams with different inputs, must reproduce their sequential results bit for bit."""
    cfg, B, T, h, w, L_txt = configs.GOLDEN["hd72_eager_split"]
    m1, m2 = _build(cfg), _build(cfg)
    with torch.no_grad():
        for p_ in m2.parameters():
            p_.mul_(1.25)
    inp1 = torch_inputs(cfg, B, T, h, w, L_txt, dtype=BF, device="cuda")
    inp2 = {k: (v if "ids" in k else (v * 0.5).to(v.dtype)) for k, v in inp1.items()}
    with torch.inference_mode():
        ref = [m1(**inp1).clone(), m2(**inp2).clone(), m1(**inp2).clone()]
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for _ in range(3):     # a few rounds so that the launches really interleave
            with torch.cuda.stream(s1):
                a = m1(**inp1)
            with torch.cuda.stream(s2):
                b = m2(**inp2)
                c = m1(**inp2)
            torch.cuda.synchronize()
            assert torch.equal(a, ref[0]) and torch.equal(b, ref[1]) and torch.equal(c, ref[2])
    ws1 = {id(v) for v in m1._osk_ws_cache.values()}
    ws2 = {id(v) for v in m2._osk_ws_cache.values()}
    assert len(ws1) >= 2 and not (ws1 & ws2)


def test_score_bound_debug_check_on_device(hip_lib, monkeypatch):
    """the bounded attention body trusts the caller's score bound.  With the debug switch on (OSK_CHECK_SCORE_BOUND=1 sets
    open_sora_amd._C.CHECK_SCORE_BOUND) every bounded call first checks the promise on the device: a golden model's forward passes
    it in every block; a call whose bound is too small raises instead of silently losing the tail of the softmax."""
    from open_sora_amd import _C, mmdit

    monkeypatch.setattr(_C, "CHECK_SCORE_BOUND", True)
    name = "hd72_eager_split"
    cfg, B, T, h, w, L_txt = configs.GOLDEN[name]
    model = mmdit.Flux(device_map="cuda", torch_dtype=torch.bfloat16, **cfg)
    model.load_state_dict(torch_params(cfg, dtype=torch.bfloat16, device="cuda"), strict=True)
    inp = torch_inputs(cfg, B, T, h, w, L_txt, dtype=torch.bfloat16, device="cuda")
    with torch.inference_mode():
        out = model(**inp)                                      # every block's bound holds
    assert torch.isfinite(out.float()).all()
    H, hd, L = 2, 72, 256
    q = torch.randn(1, L, H * hd, device="cuda").to(torch.bfloat16)
    k = torch.randn(1, L, H * hd, device="cuda").to(torch.bfloat16)
    vt = torch.zeros(1, H, hd, L, dtype=torch.bfloat16, device="cuda")
    o = torch.empty_like(q)
    with pytest.raises(RuntimeError, match="score bound"):
        _C.attention_fwd(q, k, vt, o, H, hd, hd ** -0.5, score_bound=1.0)     # |q||k| scale log2e ~ 12 >> 1
