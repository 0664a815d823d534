"""Checkpoint loading (open_sora_amd/ckpt.py, mirror of opensora/utils/ckpt.py:84-140) and the RoPE-convention weight
transform (mmdit/math.py:68-117 index maps; docs/train.md:112 "fused-rope" checkpoint), on CPU with the fp32 oracle."""
import os

import pytest
import torch

from open_sora_amd import ckpt, mmdit
from oracle import configs, mmdit_oracle as O
from tests.util import torch_inputs, torch_params


def test_rope_index_maps_are_the_reference_ones_and_inverse():
    hd = 8
    to_half = ckpt.rope_feature_index(hd, "half")
    assert to_half.tolist() == [0, 2, 4, 6, 1, 3, 5, 7]            # rearrange_tensor: 2d -> d, 2d+1 -> D/2 + d
    to_int = ckpt.rope_feature_index(hd, "interleaved")
    assert to_int.tolist() == [0, 4, 1, 5, 2, 6, 3, 7]             # reverse_rearrange_tensor
    x = torch.arange(hd)
    assert torch.equal(x[to_half][to_int], x)
    with pytest.raises(ValueError):
        ckpt.rope_feature_index(7)


@pytest.mark.parametrize("name", ["hd72_eager_split", "hd64_eager_fused", "hd128_eager_fused"])
def test_converted_weights_with_the_other_rope_convention_are_the_same_function(name):
    """eager-convention weights permuted by rearrange_tensor's map + use_liger_rope=True == the original model."""
    cfg, B, T, h, w, L_txt = configs.GOLDEN[name]
    assert not cfg["use_liger_rope"]
    sd = {k: v.double() for k, v in torch_params(cfg).items()}
    inp = {k: (v.double() if v.is_floating_point() else v) for k, v in torch_inputs(cfg, B, T, h, w, L_txt).items()}
    ref = O.forward(sd, cfg, **inp)
    sd_half = ckpt.convert_rope_convention(sd, cfg["hidden_size"], cfg["num_heads"], to="half")
    cfg_half = dict(cfg, use_liger_rope=True)
    out = O.forward(sd_half, cfg_half, **inp)
    # the liger path takes its angles in fp32 (math.py:39-47), the eager one in fp64: agreement to fp32 angle precision
    assert (out - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    # and back
    sd_back = ckpt.convert_rope_convention(sd_half, cfg["hidden_size"], cfg["num_heads"], to="interleaved")
    assert all(torch.equal(sd_back[k], sd[k]) for k in sd)
    # an untouched tensor really is untouched, a q projection really is permuted
    some_v = [k for k in sd if k.endswith("v_proj.weight") or k.endswith("img_mlp.0.weight")]
    assert all(torch.equal(sd_half[k], sd[k]) for k in some_v)
    changed = [k for k in sd if not torch.equal(sd_half[k], sd[k])]
    assert changed and all(("q_proj" in k or "k_proj" in k or "qkv" in k or "linear1" in k or "norm." in k) for k in changed)


def test_load_checkpoint_safetensors_pt_and_rename(tmp_path):
    from safetensors.torch import save_file

    cfg, *_ = configs.GOLDEN["hd64_liger_split"]
    sd = torch_params(cfg, dtype=torch.bfloat16)
    path = os.path.join(tmp_path, "model.safetensors")
    save_file({k: v.contiguous() for k, v in sd.items()}, path)
    model = mmdit.Flux(from_pretrained=path, device_map="cpu", torch_dtype=torch.bfloat16, strict_load=True, **cfg)
    got = model.state_dict()
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    # .pt file
    ppt = os.path.join(tmp_path, "model.pt")
    torch.save(sd, ppt)
    m2 = mmdit.Flux(device_map="cpu", torch_dtype=torch.bfloat16, **cfg)
    ckpt.load_checkpoint(m2, ppt, strict=True)
    assert all(torch.equal(m2.state_dict()[k], sd[k]) for k in sd)
    # rename_keys: checkpoint written under another prefix (ckpt.py:117-128)
    pren = os.path.join(tmp_path, "renamed.safetensors")
    save_file({k.replace("double_blocks.", "dbl."): v.contiguous() for k, v in sd.items()}, pren)
    m3 = mmdit.Flux(device_map="cpu", torch_dtype=torch.bfloat16, **cfg)
    ckpt.load_checkpoint(m3, pren, strict=True, rename_keys={"dbl.": "double_blocks."})
    assert all(torch.equal(m3.state_dict()[k], sd[k]) for k in sd)
    # missing file / unsupported form
    with pytest.raises(FileNotFoundError):
        ckpt.load_checkpoint(m3, os.path.join(tmp_path, "nope.safetensors"))
    with pytest.raises(ValueError):
        ckpt.load_checkpoint(m3, str(tmp_path))
