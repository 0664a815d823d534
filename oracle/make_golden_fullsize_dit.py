"""TEST INFRASTRUCTURE ONLY -- the full-size fixture of BASELINE configs[1], the configuration bench.py times, generated ONCE offline.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_fullsize_dit      (build container: /root/reference mounted; 8 cores: ~25 min, ~30 GB)

The reference's own MMDiTModel (opensora/models/mmdit/model.py:208-233 via oracle/ref_loader.py, CPU) at the XL geometry
(hidden 1152, 16 x 72, 9 + 19 blocks) runs ONE forward at B = 1, 16 x 32 x 32 = 16,384 image tokens + 512 text tokens -- the shape
bench.py pushes through `I2VDenoiser.denoise` (there as the CFG triple, B = 3) -- twice:
  * fp32: the truth;
  * bf16 parameters and activations: the reference's own eager-bf16 behaviour = the reference-precision comparator of SURVEY 8(d).
Weights: synth.make_params(mmdit_param_shapes(XL), seed 0) (bit-identical on the GPU box); inputs: synth.mmdit_inputs(XL, 1, 16, 32, 32, 512).
Stored in tests/golden/mmdit_fullsize_xl.npz:
  out_s8         truth on the token lattice: every 8th image token (offset 3), all 64 channels  [1, 2048, 64]
  ch_mean/ch_sq  per-channel mean and mean of squares of the WHOLE truth (f64 accumulation)  [64]
  e_ref          relL2(bf16 run, truth) over the whole output;  e_ref_s8: the same on the lattice
  a_ref          max |bf16 run - truth| over the whole output;  out_absmax: max |truth|
tests/test_gpu_baseline_geometry.py::test_xl_timed_configuration_vs_reference_fixture compares the HIP forward (B = 3: the timed
batch, every batch entry) with these under SURVEY 8(d)'s rule  e_ours <= max(1.5 e_ref, 2^-8).
"""
from __future__ import annotations

import os
import sys
import time

sys.dont_write_bytecode = True

import numpy as np
import torch

from . import ref_loader, synth
from .make_golden import OUT_DIR

GEOM = dict(B=1, T=16, h=32, w=32, L_txt=512)
STRIDE, OFFSET = 8, 3
# Round 6: a second fixture -- the SHIPPED geometry (11B: hidden 3072, 24 x 128, unfused q / k / v projections, Liger RoPE;
# /root/reference/configs/diffusion/inference/256px.py:36-55) at the reference's own shipped 256 px shape (129 frames of 224 x 288 px
# -> latent 33 x 28 x 36 -> 33 x 14 x 18 = 8,316 image tokens + 512 text tokens: 8,828 = 137 x 64 + 60, a ragged last key tile), at
# REDUCED DEPTH 2 + 4 (the full 19 + 38 is 11 G parameters: 44 GB in fp32, beyond this container): every kernel of the hd-128 path at
# its real width and at the real token count, six blocks deep.
#     PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_fullsize_dit 11b        (~15 min on 8 cores, ~20 GB)
GEOM_11B = dict(B=1, T=33, h=14, w=18, L_txt=512)


def xl_cfg() -> dict:
    # the XL row of open_sora_amd/configs.py restated here: oracle/ never imports the product package
    return dict(in_channels=64, vec_in_dim=768, context_in_dim=4096, mlp_ratio=4.0, theta=10000, qkv_bias=True,
                guidance_embed=False, cond_embed=True, fused_qkv=True, use_liger_rope=False,
                hidden_size=1152, num_heads=16, depth=9, depth_single_blocks=19, axes_dim=[8, 32, 32])


def cfg_11b_d2s4() -> dict:
    # the 11B row of open_sora_amd/configs.py at depth 2 + 4, restated (oracle/ never imports the product package)
    return dict(in_channels=64, vec_in_dim=768, context_in_dim=4096, mlp_ratio=4.0, theta=10000, qkv_bias=True,
                guidance_embed=False, cond_embed=True, fused_qkv=False, use_liger_rope=True,
                hidden_size=3072, num_heads=24, depth=2, depth_single_blocks=4, axes_dim=[16, 56, 56])


def summarize(out: torch.Tensor) -> dict:
    """the stored view of a [1, L_img, 64] prediction (also what the test computes from each batch entry of the HIP output)"""
    d = out.double()
    return {"out_s8": out[:, OFFSET::STRIDE].float().numpy().copy(),
            "ch_mean": d.mean(dim=(0, 1)).numpy().copy(), "ch_sq": (d * d).mean(dim=(0, 1)).numpy().copy()}


def rel_l2(a, b) -> float:
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def main(which: str = "xl"):
    torch.set_num_threads(os.cpu_count() or 1)
    t0 = time.time()
    cfg, GEOM, name = (xl_cfg(), globals()["GEOM"], "mmdit_fullsize_xl") if which == "xl" else (cfg_11b_d2s4(), GEOM_11B, "mmdit_fullsize_11b_d2s4")
    M, _, _ = ref_loader.mmdit()
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(synth.mmdit_param_shapes(cfg), 0, workers=os.cpu_count()).items()}
    print(f"weights: {time.time() - t0:.0f} s ({sum(v.numel() for v in sd.values()) / 1e9:.2f} G parameters)", flush=True)
    inp = {k: torch.from_numpy(v) for k, v in synth.mmdit_inputs(cfg, GEOM["B"], GEOM["T"], GEOM["h"], GEOM["w"], GEOM["L_txt"]).items()}
    model = M.Flux(device_map="cpu", torch_dtype=torch.float32, **cfg)
    model.load_state_dict(sd, strict=True)
    del sd
    with torch.inference_mode():
        truth = model(**inp)
    print(f"fp32 forward: {time.time() - t0:.0f} s  out {tuple(truth.shape)} |out|max {float(truth.abs().max()):.4f}", flush=True)
    out = summarize(truth)
    np.savez_compressed(os.path.join(OUT_DIR, name + ".partial.npz"), **out)
    model = model.to(torch.bfloat16)
    inp16 = {k: (v if "ids" in k else v.bfloat16()) for k, v in inp.items()}
    with torch.inference_mode():
        ref = model(**inp16).float()
    print(f"bf16 forward: {time.time() - t0:.0f} s  finite {bool(torch.isfinite(ref).all())}", flush=True)
    out["e_ref"] = np.float64(rel_l2(ref, truth))
    out["e_ref_s8"] = np.float64(rel_l2(ref[:, OFFSET::STRIDE], truth[:, OFFSET::STRIDE]))
    out["a_ref"] = np.float64((ref.double() - truth.double()).abs().max())
    out["out_absmax"] = np.float64(truth.abs().max())
    print({k: float(out[k]) for k in ("e_ref", "e_ref_s8", "a_ref", "out_absmax")}, flush=True)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    os.remove(os.path.join(OUT_DIR, name + ".partial.npz"))
    print(f"{path}: {os.path.getsize(path)} B")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "xl")
