"""TEST INFRASTRUCTURE ONLY -- the full-size fixture of BASELINE configs[2] (VERDICT r3 weak #1b), generated ONCE offline.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_fullsize        (build container: /root/reference mounted; ~20 min, ~25 GB)

The reference's own AutoencoderKLCausal3D (fp32, CPU, oracle/ref_loader.py) at the shipped widths (128/256/512/512, 2 layers per
block) encodes synth.vae_video(1, 33, 256, 256) and decodes synth.vae_latent(1, 9, 32, 32) -- the tensors bench.py's `vae`
sub-object pushes through the HIP path.  Stored in tests/golden/vae_fullsize_cfg3.npz (weights and inputs are regenerated from
oracle/synth.py, never stored):
  z        the whole latent mean [1, 16, 9, 32, 32] (147 k values)
  dec_s8   the decoded video sampled at every 8th row and column (offset 3), all 33 frames: [1, 3, 33, 32, 32]
  dec_mean / dec_sq   per (channel, frame) mean and mean of squares of the whole decoded video (f64 accumulation): [3, 33]
  e_ref_z / e_ref_dec   relL2 of the reference's own eager-bf16 run (bf16 parameters and activations) against the fp32 run, over the
                        whole latent / the whole decoded video; a_ref_z / a_ref_dec the max-abs differences; z_absmax / dec_absmax
                        (round 5: SURVEY 8(d)'s reference-precision comparator, committed with the fixture so that the GPU test's
                        bound is max(1.5 e_ref, 2^-8) instead of a hand-set number)
tests/test_gpu_vae.py::test_full_size_encode_decode_vs_reference_fixture compares the HIP path with these.
"""
from __future__ import annotations

import os
import sys
import time

sys.dont_write_bytecode = True

import numpy as np
import torch

from . import configs, synth
from .make_golden import OUT_DIR, reference_vae

CFG = dict(configs._VAE, block_out_channels=(128, 256, 512, 512), layers_per_block=2)
SHAPE = (1, 33, 256, 256)
STRIDE, OFFSET = 8, 3


def summarize_dec(dec: torch.Tensor) -> dict:
    """the stored view of a decoded [1, 3, T, H, W] video (also what the test computes from the HIP output)"""
    d = dec.double()
    return {"dec_s8": dec[:, :, :, OFFSET::STRIDE, OFFSET::STRIDE].float().numpy().copy(),
            "dec_mean": d.mean(dim=(3, 4))[0].numpy().copy(), "dec_sq": (d * d).mean(dim=(3, 4))[0].numpy().copy()}


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    t0 = time.time()
    model = reference_vae(CFG)
    print(f"weights: {time.time() - t0:.0f} s", flush=True)
    B, T, H, W = SHAPE
    x = torch.from_numpy(synth.vae_video(B, T, H, W))
    out = {}
    with torch.inference_mode():
        z = model.encode(x, sample_posterior=False)
        out["z"] = z.float().numpy().copy()
        print(f"encode: {time.time() - t0:.0f} s  z {tuple(z.shape)}", flush=True)
        zin = torch.from_numpy(synth.vae_latent(B, z.shape[2], z.shape[3], z.shape[4]))
        dec = model.decode(zin)
        print(f"decode: {time.time() - t0:.0f} s  dec {tuple(dec.shape)}", flush=True)
        out.update(summarize_dec(dec))
        # the reference-precision comparator: the same module with bf16 parameters on bf16 inputs
        m16 = model.to(torch.bfloat16)
        z16 = m16.encode(x.bfloat16(), sample_posterior=False).float()
        print(f"bf16 encode: {time.time() - t0:.0f} s  finite {bool(torch.isfinite(z16).all())}", flush=True)
        d16 = m16.decode(zin.bfloat16()).float()
        print(f"bf16 decode: {time.time() - t0:.0f} s  finite {bool(torch.isfinite(d16).all())}", flush=True)
        for tag, a, b in (("z", z16, z.float()), ("dec", d16, dec.float())):
            out["e_ref_" + tag] = np.float64((a.double() - b.double()).norm() / b.double().norm())
            out["a_ref_" + tag] = np.float64((a.double() - b.double()).abs().max())
            out[tag + "_absmax"] = np.float64(b.abs().max())
        print({k: float(v) for k, v in out.items() if np.ndim(v) == 0}, flush=True)
    path = os.path.join(OUT_DIR, "vae_fullsize_cfg3.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {os.path.getsize(path)} B")


if __name__ == "__main__":
    main()
