#!/usr/bin/env python
"""Per-rank kernel times of BASELINE configs[4] (11B geometry, 64 x 720p latent read as T_lat = 64: L = 230,912 tokens, SP = 8; bf16 and
the fp8 mode) measured on ONE GPU, with the loop bodies the model selects there (bounded FAST body in bf16, the fp8 P.V body in fp8 mode)
-- a thin wrapper of tools/rank_shapes.py (round 6: the earlier version timed the general body; see microbench_cfg4.py)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import rank_shapes

print(json.dumps(rank_shapes.measure(torch.device("cuda", 0), ("cfg4",))["cfg4"]), flush=True)
