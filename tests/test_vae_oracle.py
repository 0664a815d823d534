"""CPU: the VAE oracle (oracle/vae_oracle.py) pinned (a) against the goldens produced by the REAL reference
AutoencoderKLCausal3D (oracle/make_golden.py) — runs everywhere — and (b) against the reference imported from
/root/reference, incl. the state-dict key set and bf16 behaviour (build container only)."""
import os

import numpy as np
import pytest
import torch

from oracle import configs, ref_loader, synth, vae_oracle as V

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sd(cfg, dtype=torch.float32):
    return {k: torch.from_numpy(v).to(dtype) for k, v in synth.make_params(synth.vae_param_shapes(cfg), 0).items()}


def _close(a, g, tol=3e-5):
    return np.abs(a - g).max() <= tol * max(1.0, float(np.abs(g).max()))


@pytest.mark.parametrize("name", list(configs.VAE_GOLDEN))
def test_vae_oracle_matches_reference_golden(name):
    cfg, B, T, H, W = configs.VAE_GOLDEN[name]
    g = np.load(os.path.join(GOLDEN_DIR, f"vae_{name}.npz"))
    sd = _sd(cfg)
    x = torch.from_numpy(synth.vae_video(B, T, H, W))
    with torch.inference_mode():
        mom = V.encode_moments(sd, cfg, x)
        z = V.encode(sd, cfg, x)
        zin = torch.from_numpy(synth.vae_latent(B, *g["z"].shape[2:]))
        dec = V.decode(sd, cfg, zin)
    assert z.shape == g["z"].shape and dec.shape == g["dec"].shape
    assert _close(z.numpy(), g["z"])
    assert _close(mom.chunk(2, 1)[1].clamp(-30, 20).numpy(), g["logvar"])
    assert _close(dec.numpy(), g["dec"])


@pytest.mark.parametrize("name", list(configs.VAE_TILED_GOLDEN))
def test_vae_oracle_tiling_matches_reference_golden(name):
    cfg, B, T, H, W = configs.VAE_TILED_GOLDEN[name]
    g = np.load(os.path.join(GOLDEN_DIR, f"vae_{name}.npz"))
    sd = _sd(cfg)
    x = torch.from_numpy(synth.vae_video(B, T, H, W))
    with torch.inference_mode():
        z = V.encode_tiled(sd, cfg, x)
        zin = torch.from_numpy(synth.vae_latent(B, *g["z"].shape[2:]))
        dec = V.decode_tiled(sd, cfg, zin)
    assert z.shape == g["z"].shape and dec.shape == g["dec"].shape
    assert _close(z.numpy(), g["z"])
    assert _close(dec.numpy(), g["dec"])


def test_vae_flops_match_survey():
    """[1,3,33,256,256] -> encode 2.00e13, decode 3.73e13."""
    cfg = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=16)
    enc, dec = V.conv_flops(cfg, 33, 256, 256)
    assert abs(enc / 2.00e13 - 1) < 0.02 and abs(dec / 3.73e13 - 1) < 0.02, (enc, dec)


@pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not mounted")
def test_vae_oracle_vs_reference_module_fp32_and_bf16():
    from oracle.make_golden import reference_vae

    cfg, B, T, H, W = configs.VAE_GOLDEN["c32_lpb1"]
    model = reference_vae(cfg)
    sd = _sd(cfg)
    assert set(model.state_dict().keys()) == set(sd.keys())
    x = torch.from_numpy(synth.vae_video(B, T, H, W))
    with torch.inference_mode():
        z_ref = model.encode(x, sample_posterior=False)
        d_ref = model.decode(z_ref)
        z = V.encode(sd, cfg, x)
        d = V.decode(sd, cfg, z_ref)
        assert _close(z.numpy(), z_ref.numpy()) and _close(d.numpy(), d_ref.numpy())
        # bf16: the oracle run with bf16 tensors sits at the same distance from fp32 as the reference in bf16
        mb = model.to(torch.bfloat16)
        sdb = _sd(cfg, torch.bfloat16)
        zb_ref = mb.encode(x.bfloat16(), sample_posterior=False).float()
        zb = V.encode(sdb, cfg, x.bfloat16()).float()
        e_ref = float((zb_ref - z_ref).norm() / z_ref.norm())
        e_mine = float((zb - z_ref).norm() / z_ref.norm())
        assert e_mine <= 1.5 * e_ref and e_ref <= 1.5 * e_mine, (e_ref, e_mine)
