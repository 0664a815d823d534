#!/usr/bin/env python
"""relL2 of the fp8 mode against the bf16 mode on one full denoise forward (random-init weights, synthetic inputs):
the whole-model quantisation error of MMDiTModel.enable_fp8() (gate 5e-2 for the fp8 configuration)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_sora_amd import configs, mmdit, sampling

dev = torch.device("cuda", 0)
for name, T, hw, nb in (("XL", 16, 64, 1), ("11B", 4, 32, 1)):
    cfg = dict(configs.MMDIT[name])
    torch.manual_seed(1234)
    model = mmdit.Flux(device_map=dev, torch_dtype=torch.bfloat16, **cfg)
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if n_.startswith("cond_in"):
                p_.normal_(0, 0.02)
    L_img, L_txt = T * (hw // 2) ** 2, 512
    g = torch.Generator(device=dev).manual_seed(42)
    img = torch.randn(nb, L_img, 64, device=dev, generator=g).to(torch.bfloat16)
    txt = (torch.randn(nb, L_txt, cfg["context_in_dim"], device=dev, generator=g) * 0.2).to(torch.bfloat16)
    y_vec = torch.randn(nb, cfg["vec_in_dim"], device=dev, generator=g).to(torch.bfloat16)
    img_ids, txt_ids = sampling.prepare_ids(nb, T, hw, hw, L_txt, dev, torch.bfloat16)
    cond = torch.zeros(nb, L_img, 68, device=dev, dtype=torch.bfloat16)
    t = torch.full((nb,), 0.7, dtype=torch.bfloat16, device=dev)
    kw = dict(img=img, img_ids=img_ids, txt=txt, txt_ids=txt_ids, timesteps=t, y_vec=y_vec, cond=cond)
    with torch.inference_mode():
        ref = model(**kw).float()
        out = model.enable_fp8()(**kw).float()
    rel = ((out - ref).norm() / ref.norm()).item()
    print(json.dumps({"model": name, "tokens": L_img + L_txt, "relL2_fp8_vs_bf16": round(rel, 5),
                      "max_abs": round((out - ref).abs().max().item(), 4), "ref_rms": round(ref.pow(2).mean().sqrt().item(), 4)}), flush=True)
    del model
    torch.cuda.empty_cache()
