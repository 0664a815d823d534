"""CPU: the oracle restatement (oracle/mmdit_oracle.py) against the committed goldens, which were produced
by the REAL reference Python (oracle/make_golden.py).  This is what pins the oracle on machines without
/root/reference (e.g. the GPU box)."""
import os

import numpy as np
import pytest
import torch

from oracle import configs, mmdit_oracle as O
from tests.util import torch_inputs, torch_params

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", list(configs.GOLDEN))
def test_oracle_matches_reference_golden(name):
    cfg, B, T, h, w, L_txt = configs.GOLDEN[name]
    g = np.load(os.path.join(GOLDEN_DIR, f"mmdit_{name}.npz"))
    sd = torch_params(cfg)
    inp = torch_inputs(cfg, B, T, h, w, L_txt)
    taps = {}
    with torch.inference_mode():
        out = O.forward(sd, cfg, **inp, taps=taps)
    scale = float(np.abs(g["out"]).max())
    assert np.abs(out.numpy() - g["out"]).max() <= 2e-5 * max(scale, 1.0)
    assert np.abs(taps["vec"].numpy() - g["vec"]).max() <= 2e-5 * max(float(np.abs(g["vec"]).max()), 1.0)
    assert np.abs(taps["double.0.img"].numpy() - g["double0_img"]).max() <= 2e-5 * max(float(np.abs(g["double0_img"]).max()), 1.0)
    assert np.abs(taps["double.0.txt"].numpy() - g["double0_txt"]).max() <= 2e-5 * max(float(np.abs(g["double0_txt"]).max()), 1.0)


def test_synth_is_stable():
    """The synthetic generator is a pure function of (name, seed, shape): a few pinned values."""
    from oracle import synth

    a = synth.normal("x", 0, (4,))
    np.testing.assert_allclose(a[:3], np.array([-1.4868279, -0.43632686, -0.28022543], np.float32), rtol=0, atol=1e-6)
    r = synth.bf16_round(np.array([1.00390625, 1.01171875, -0.1], np.float32))
    assert r[0] == 1.0 and r[1] == np.float32(1.015625)


def test_flops_formula_matches_survey():
    """XL, B=1, L=16896 -> 5.19e13; 11B 256px B=3 -> 5.06e14."""
    xl = O.flops_per_forward(configs.MMDIT["XL"], 1, 16384, 512)
    assert abs(xl / 5.19e13 - 1) < 0.01
    big = O.flops_per_forward(configs.MMDIT["11B"], 3, 8316, 512)
    assert abs(big / 5.06e14 - 1) < 0.01
