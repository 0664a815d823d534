"""TEST INFRASTRUCTURE ONLY — generate tests/golden/*.npz from the REAL reference Python.

Run in the build container (where /root/reference is mounted):
    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden
For every entry of oracle.configs.GOLDEN it instantiates the reference's own MMDiTModel
(opensora/models/mmdit/model.py via oracle/ref_loader.py, fp32, CPU), loads the deterministic synthetic
weights of oracle/synth.py, runs forward() on the synthetic inputs and stores the output plus a few
intermediates.  Weights and inputs are NOT stored: they are regenerated bit-identically from (name, seed).
"""
from __future__ import annotations

import os
import sys

sys.dont_write_bytecode = True

import numpy as np
import torch

from . import configs, ref_loader, synth

OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def reference_mmdit_outputs(cfg, B, T, h, w, L_txt, seed=0):
    M, layers, _ = ref_loader.mmdit()
    model = M.Flux(device_map="cpu", torch_dtype=torch.float32, **cfg)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(synth.mmdit_param_shapes(cfg), seed).items()}
    model.load_state_dict(sd, strict=True)
    inp = {k: torch.from_numpy(v) for k, v in synth.mmdit_inputs(cfg, B, T, h, w, L_txt).items()}
    taps = {}
    with torch.inference_mode():
        img, txt, vec, pe = model.prepare_block_inputs(**inp)
        taps["vec"] = vec.numpy().copy()
        img1, txt1 = model.double_blocks[0](img, txt, vec, pe)
        taps["double0_img"] = img1.numpy().copy()
        taps["double0_txt"] = txt1.numpy().copy()
        out = model(**inp)
    taps["out"] = out.numpy().copy()
    return taps


def reference_vae(cfg):
    """The reference's own AutoencoderKLCausal3D (fp32, CPU) with the synthetic weights of oracle/synth.py."""
    ae = ref_loader.hunyuan_ae()
    model = ae.AutoencoderKLCausal3D(ae.AutoEncoder3DConfig(from_pretrained=None, **cfg))
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(synth.vae_param_shapes(cfg), 0).items()}
    model.load_state_dict(sd, strict=True)
    return model.eval()


def reference_vae_outputs(cfg, B, T, H, W, tiled=False):
    model = reference_vae(cfg)
    if tiled:
        model.enable_tiling()
    x = torch.from_numpy(synth.vae_video(B, T, H, W))
    taps = {}
    with torch.inference_mode():
        z, post = model.encode(x, sample_posterior=False, return_posterior=True)
        taps["z"] = z.numpy().copy()
        taps["logvar"] = post.logvar.numpy().copy()
        zin = torch.from_numpy(synth.vae_latent(B, z.shape[2], z.shape[3], z.shape[4]))
        taps["dec"] = model.decode(zin).numpy().copy()
    return taps


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    for table, tiled in ((configs.VAE_GOLDEN, False), (configs.VAE_TILED_GOLDEN, True)):
        for name, (cfg, B, T, H, W) in table.items():
            taps = reference_vae_outputs(cfg, B, T, H, W, tiled)
            path = os.path.join(OUT_DIR, f"vae_{name}.npz")
            np.savez_compressed(path, **{k: v.astype(np.float32) for k, v in taps.items()})
            print(f"{path}: z {taps['z'].shape} dec {taps['dec'].shape} ({os.path.getsize(path)} B)")
    if "--vae-only" in sys.argv:
        return
    for name, (cfg, B, T, h, w, L_txt) in configs.GOLDEN.items():
        taps = reference_mmdit_outputs(cfg, B, T, h, w, L_txt)
        path = os.path.join(OUT_DIR, f"mmdit_{name}.npz")
        np.savez_compressed(path, **{k: v.astype(np.float32) for k, v in taps.items()})
        print(f"{path}: out {taps['out'].shape} |out|max {np.abs(taps['out']).max():.4f} ({os.path.getsize(path)} B)")


if __name__ == "__main__":
    main()
