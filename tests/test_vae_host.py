"""CPU: host orchestration of open_sora_amd.hunyuan_vae (NDHWC engine, weight re-layout [Cout][tap*Cin], channel
padding, fused upsample / residual flags, mid-block attention on GEMMs, tiling + blending, API mirror) driven through
the CPU emulation of the kernels' semantics (tests/cpu_ops.py) and compared with the goldens made by the REAL
reference.  The kernels themselves are checked on the GPU by tests/test_gpu_vae.py."""
import os

import numpy as np
import pytest
import torch

from oracle import configs, synth, vae_oracle as V
from tests import cpu_ops
from tests.util import assert_parity, finite_retry

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BF = torch.bfloat16


@pytest.fixture()
def cpu_vae(hip_lib):
    from open_sora_amd import hunyuan_vae, mmdit

    mmdit.set_ops_for_testing(cpu_ops)
    yield hunyuan_vae
    mmdit.set_ops_for_testing(hip_lib)


def _sd(cfg, dtype=torch.float32):
    return {k: torch.from_numpy(v).to(dtype) for k, v in synth.make_params(synth.vae_param_shapes(cfg), 0).items()}


def _model(vae, cfg):
    m = vae.CausalVAE3D_HUNYUAN(device_map="cpu", torch_dtype=BF, **cfg)
    m.load_state_dict(_sd(cfg, BF), strict=True)
    return m


@pytest.mark.parametrize("name", list(configs.VAE_GOLDEN))
def test_vae_engine_vs_golden(cpu_vae, name):
    cfg, B, T, H, W = configs.VAE_GOLDEN[name]
    g = np.load(os.path.join(GOLDEN_DIR, f"vae_{name}.npz"))
    m = _model(cpu_vae, cfg)
    x = torch.from_numpy(synth.vae_video(B, T, H, W))
    zin = torch.from_numpy(synth.vae_latent(B, *g["z"].shape[2:]))
    sdb = _sd(cfg, BF)
    with torch.inference_mode():
        z = m.encode(x.to(BF), sample_posterior=False)
        dec = m.decode(zin.to(BF))
        z_ref = finite_retry(lambda: V.encode(sdb, cfg, x.to(BF)))
        d_ref = finite_retry(lambda: V.decode(sdb, cfg, zin.to(BF)))
    assert list(z.shape) == list(g["z"].shape) and list(dec.shape) == list(g["dec"].shape)
    assert m.get_latent_size([T, H, W]) == list(g["z"].shape[2:])
    assert_parity(z, torch.from_numpy(g["z"]), z_ref, f"vae encode host [{name}]")
    assert_parity(dec, torch.from_numpy(g["dec"]), d_ref, f"vae decode host [{name}]")


def test_vae_tiling_vs_golden(cpu_vae):
    name = "c32_tiled"
    cfg, B, T, H, W = configs.VAE_TILED_GOLDEN[name]
    g = np.load(os.path.join(GOLDEN_DIR, f"vae_{name}.npz"))
    m = _model(cpu_vae, cfg)
    m.enable_tiling()
    x = torch.from_numpy(synth.vae_video(B, T, H, W))
    zin = torch.from_numpy(synth.vae_latent(B, *g["z"].shape[2:]))
    sdb = _sd(cfg, BF)
    with torch.inference_mode():
        z = m.encode(x.to(BF), sample_posterior=False)
        dec = m.decode(zin.to(BF))
        z_ref = finite_retry(lambda: V.encode_tiled(sdb, cfg, x.to(BF)))
        d_ref = finite_retry(lambda: V.decode_tiled(sdb, cfg, zin.to(BF)))
    assert_parity(z, torch.from_numpy(g["z"]), z_ref, "vae tiled encode host")
    assert_parity(dec, torch.from_numpy(g["dec"]), d_ref, "vae tiled decode host")


def test_vae_api_mirror(cpu_vae):
    cfg, B, T, H, W = configs.VAE_GOLDEN["c32_single_frame"]
    m = _model(cpu_vae, cfg)
    x = torch.from_numpy(synth.vae_video(B, T, H, W)).to(BF)
    with torch.inference_mode():
        gen = torch.Generator().manual_seed(1)
        z, post = m.encode(x, sample_posterior=True, return_posterior=True, generator=gen)
        assert z.shape == post.mean.shape and post.logvar.min() >= -30 and post.logvar.max() <= 20
        dec, post2, z2 = m(x, sample_posterior=False)
        assert dec.shape == x.shape and torch.equal(z2, m.scale_factor * post2.mode())
        with pytest.raises(AssertionError):
            m.encode(x[0])
        with pytest.raises(RuntimeError):
            m.encoder(x)  # parameter holders have no eager arithmetic


def test_vae_built_and_loaded_inside_inference_mode(cpu_vae):
    """scripts/vae/inference.py and scripts/diffusion/inference.py build the VAE inside torch.inference_mode(): its
    parameters are inference tensors (no version counter); the per-layer plan key must not read `_version` on them."""
    name = "c32_lpb1"
    cfg, B, T, H, W = configs.VAE_GOLDEN[name]
    x = torch.from_numpy(synth.vae_video(B, T, H, W)).to(BF)
    outside = _model(cpu_vae, cfg)
    with torch.inference_mode():
        inside = _model(cpu_vae, cfg)
        assert all(p.is_inference() for p in inside.parameters())
        za = inside.encode(x, sample_posterior=False)
        zb = outside.encode(x, sample_posterior=False)
        da = inside.decode(za)
    assert torch.equal(za, zb) and torch.isfinite(da.float()).all()


def test_vae_plans_do_not_travel_and_follow_state_loads(cpu_vae):
    """the per-layer weight images live in a weak map (not on the plain torch layers), so copy.deepcopy / pickle of the
    VAE carry none of them; a load_state_dict on a SUB-module under inference_mode (no version counter to key on) drops the
    layer's image through its post hook."""
    import copy
    import pickle

    from open_sora_amd import hunyuan_vae

    cfg, B, T, H, W = configs.VAE_GOLDEN["c32_lpb1"]
    x = torch.from_numpy(synth.vae_video(B, T, H, W)).to(BF)
    with torch.inference_mode():
        m = _model(cpu_vae, cfg)
        z0 = m.encode(x, sample_posterior=False)
        n_plans = sum(1 for mod in m.modules() if mod in hunyuan_vae._PLANS)
        assert n_plans > 10
        assert not any(k.startswith("_osk_plan") for mod in m.modules() for k in mod.__dict__)
        clone = copy.deepcopy(m)
        assert sum(1 for mod in clone.modules() if mod in hunyuan_vae._PLANS) == 0
        assert len(pickle.dumps(m.encoder.conv_in)) < 4 * sum(p.numel() * p.element_size() for p in m.encoder.conv_in.parameters()) + 4096
        assert torch.equal(clone.encode(x, sample_posterior=False), z0)
        conv = m.encoder.conv_in
        conv.load_state_dict({k: (v * 2.0).to(v.dtype) for k, v in conv.state_dict().items()})
        assert conv.conv not in hunyuan_vae._PLANS
        z1 = m.encode(x, sample_posterior=False)
    assert not torch.equal(z1, z0)


def test_groupnorm_fold_is_opt_in_and_equivalent(cpu_vae):
    """hunyuan_vae.FOLD_GN (default off: equal encode + decode time on the 1.4 kW board): norm -> SiLU -> conv through
    osk_groupnorm_table_f32 + osk_causal_conv3d_gnin_ndhwc_bf16 gives the resnet block's output of the apply + conv path, takes
    the fused output statistics along, and falls back where the sliding-window kernels do not take the shape"""
    vae = cpu_vae
    assert vae.FOLD_GN is False
    torch.manual_seed(0)
    blk = vae.ResnetBlockCausal3D(in_channels=128, out_channels=128).to(BF)
    x = torch.randn(1, 3, 16, 16, 128).to(BF)
    calls = []
    orig = cpu_ops.causal_conv3d_gn_in

    def spy(*a, **k):
        r = orig(*a, **k)
        calls.append(r)
        return r

    cpu_ops.causal_conv3d_gn_in = spy
    try:
        with torch.inference_mode():
            plain = vae._resnet(blk, x)
            assert not calls
            vae.FOLD_GN = True
            fold = vae._resnet(blk, x)
            assert calls == [(True, True), (True, True)]           # both convs folded, both with fused output statistics
            assert getattr(fold, "_osk_gn", None) is not None
            small = vae._resnet(blk, x[:, :1])                     # Cout == 128 needs frame pairs: declined, apply + conv runs
            assert calls[2:] == [(False, False), (False, False)] and small.shape == (1, 1, 16, 16, 128)
    finally:
        cpu_ops.causal_conv3d_gn_in = orig
        vae.FOLD_GN = False
    d = (fold.float() - plain.float()).abs().max()
    assert d <= 2.0 ** -6 * plain.float().abs().max(), d


def test_plugin_targets_run_on_the_reference_modules(cpu_vae):
    """SURVEY 8(b) "VAE plugin point" (VERDICT r4 missing #3): the reference's OWN AutoencoderKLCausal3D (oracle/ref_loader.py) with
    open_sora_amd.vae_plugin's from_native_module targets installed on its encoder / decoder -- the ShardFormer replacement
    convention of hunyuan_vae/policy.py:13-48 -- against the same module un-patched (fp32 truth, bf16 comparator).  The state dict
    is untouched, the reference's own encode / decode / tiling Python drives the kernels, uninstall restores the native modules;
    and the per-layer target (one CausalConv3d) equals the native layer."""
    from oracle import make_golden, ref_loader

    if not ref_loader.available():
        pytest.skip("/root/reference not mounted")
    from open_sora_amd import vae_plugin

    cfg, B, T, H, W = configs.VAE_GOLDEN["c32_single_frame"] if "c32_single_frame" in configs.VAE_GOLDEN else next(iter(configs.VAE_GOLDEN.values()))
    name = next(n for n, v in configs.VAE_GOLDEN.items() if v[0] is cfg)
    g = np.load(os.path.join(GOLDEN_DIR, f"vae_{name}.npz"))
    ref = make_golden.reference_vae(cfg)
    x = torch.from_numpy(synth.vae_video(B, T, H, W))
    zin = torch.from_numpy(synth.vae_latent(B, *g["z"].shape[2:]))
    with torch.inference_mode():
        z_t = ref.encode(x, sample_posterior=False)
        d_t = ref.decode(zin)
        refb = ref.to(BF)
        keys = list(refb.state_dict().keys())
        z_r = finite_retry(lambda: refb.encode(x.to(BF), sample_posterior=False))
        d_r = finite_retry(lambda: refb.decode(zin.to(BF)))
        vae_plugin.install(refb)
        assert isinstance(refb.encoder, vae_plugin.HipEncoderCausal3D) and isinstance(refb.decoder, vae_plugin.HipDecoderCausal3D)
        assert list(refb.state_dict().keys()) == keys, "the plug-in changed the state dict"
        z = refb.encode(x.to(BF), sample_posterior=False)
        dec = refb.decode(zin.to(BF))
        assert z.dtype == BF and dec.dtype == BF
        assert_parity(z, z_t, z_r, f"reference VAE + HIP encoder target [{name}]")
        assert_parity(dec, d_t, d_r, f"reference VAE + HIP decoder target [{name}]")
        # per-layer granularity: one CausalConv3d of the (native) decoder
        vae_plugin.uninstall(refb)
        assert not isinstance(refb.encoder, vae_plugin.HipEncoderCausal3D)
        conv = refb.decoder.conv_in
        h_nat = conv(zin.to(BF))
        h_hip = vae_plugin.HipCausalConv3d.from_native_module(conv)(zin.to(BF))
        assert h_hip.shape == h_nat.shape and (h_hip.float() - h_nat.float()).abs().max() <= 2.0 ** -6 * h_nat.float().abs().max() + 1e-3


def test_plugin_refuses_configurations_the_engine_does_not_implement(cpu_vae):
    """the NDHWC engine hard-codes the shipped settings (replicate padding, output_scale_factor 1, SiLU, ...); adopting a
    reference module configured otherwise would change its numerics silently -- from_native_module refuses it."""
    from oracle import ref_loader

    if not ref_loader.available():
        pytest.skip("/root/reference not mounted")
    from open_sora_amd import vae_plugin

    _, blocks = ref_loader.hunyuan_vae()
    ok = blocks.CausalConv3d(8, 8, kernel_size=3)
    vae_plugin.HipCausalConv3d.from_native_module(ok)
    with pytest.raises(ValueError, match="pad_mode"):
        vae_plugin.HipCausalConv3d.from_native_module(blocks.CausalConv3d(8, 8, kernel_size=3, pad_mode="constant"))
    rb = blocks.ResnetBlockCausal3D(in_channels=32, out_channels=32, groups=32, output_scale_factor=2.0)
    with pytest.raises(ValueError, match="output_scale_factor"):
        vae_plugin._assert_supported(rb)
