#!/usr/bin/env python
"""Per-rank kernel times of BASELINE configs[3] (XL, 51 x 720p latent, L = 184,112 tokens, SP = 8) measured on ONE GPU, with the
bounded (FAST / wide) attention body the model selects there -- a thin wrapper of tools/rank_shapes.py (round 6: the round-3 version
of this script called attention_fwd without the score bound, i.e. timed the general body a sequence-parallel rank never runs)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import rank_shapes

print(json.dumps(rank_shapes.measure(torch.device("cuda", 0), ("cfg3",))["cfg3"]), flush=True)
