"""CPU, build container only: the oracle restatement against the reference's own Python imported from
/root/reference (skipped where the reference is not mounted, e.g. on the GPU box)."""
import pytest
import torch

from oracle import configs, mmdit_oracle as O, ref_loader
from tests.util import rel_l2, torch_inputs, torch_params

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not mounted")


@pytest.mark.parametrize("name", ["hd64_eager_fused", "hd72_liger_fused", "hd128_liger_split"])
def test_forward_fp32_and_bf16(name):
    cfg, B, T, h, w, L_txt = configs.GOLDEN[name]
    M, _, _ = ref_loader.mmdit()
    model = M.Flux(device_map="cpu", torch_dtype=torch.float32, **cfg)
    sd = torch_params(cfg)
    model.load_state_dict(sd, strict=True)
    inp = torch_inputs(cfg, B, T, h, w, L_txt)
    with torch.inference_mode():
        ref = model(**inp)
        mine = O.forward(sd, cfg, **inp)
    assert (ref - mine).abs().max() <= 2e-5 * max(float(ref.abs().max()), 1.0)
    # bf16: same rounding points -> both sit at the same distance from the fp32 truth
    mb = model.to(torch.bfloat16)
    sdb = {k: v.bfloat16() for k, v in sd.items()}
    inpb = torch_inputs(cfg, B, T, h, w, L_txt, dtype=torch.bfloat16)
    with torch.inference_mode():
        refb = mb(**inpb).float()
        mineb = O.forward(sdb, cfg, **inpb).float()
    e_ref, e_mine = rel_l2(refb, ref), rel_l2(mineb, ref)
    assert e_mine <= 1.5 * e_ref and e_ref <= 1.5 * e_mine, (e_ref, e_mine)


def test_reference_error_conventions():
    """ValueError for ndim != 3 / missing cond (model.py:172-179) are mirrored by the oracle."""
    cfg, B, T, h, w, L_txt = configs.GOLDEN["hd64_eager_fused"]
    sd = torch_params(cfg)
    inp = torch_inputs(cfg, B, T, h, w, L_txt)
    bad = dict(inp)
    bad["img"] = inp["img"][0]
    with pytest.raises(ValueError):
        O.forward(sd, cfg, **bad)
    bad = dict(inp)
    bad.pop("cond")
    with pytest.raises(ValueError):
        O.forward(sd, cfg, **bad)
