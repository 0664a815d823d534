import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import _altlib
print("lib:", _altlib.install() or "shipped")
import torch
from open_sora_amd import _C
from oracle import synth
DEV, BF = "cuda", torch.bfloat16
def rnd(name, shape, std=1.0, seed=7, dtype=BF):
    return torch.from_numpy(synth.normal(name, seed, shape, std=std)).to(DEV).to(dtype)
for (B, L, N, K, gf) in [(3, 5000, 2304, 192, 1000), (3, 5000, 2304, 192, 0), (3, 5000, 2304, 192, 1024), (1, 1024, 512, 192, 0)]:
    a = rnd("a", (B, L, K), seed=31); w = rnd("w", (N, K), std=K ** -0.5, seed=32); bias = rnd("b", (N,), std=0.3, dtype=torch.float32, seed=33)
    out = torch.empty(B, L, N, dtype=BF, device=DEV)
    _C.gemm(a, w, bias, out, gelu_from=gf)
    v = (a.float().reshape(B * L, K) @ w.float().T + bias).reshape(B, L, N)
    v = torch.cat([v[..., :gf], torch.nn.functional.gelu(v[..., gf:], approximate="tanh")], -1)
    d = (out.float() - v).abs()
    bad = d > (3e-3 + 2 ** -7 * v.abs())
    idx = bad.nonzero()
    rec = {"case": [B, L, N, K, gf], "bad": int(bad.sum()), "max_err": float(d.max())}
    if len(idx):
        cols = idx[:, 2]; rows = idx[:, 0] * L + idx[:, 1]
        rec.update(col_min=int(cols.min()), col_max=int(cols.max()), cols_mod256=sorted(set((cols % 256).tolist()))[:12], ncols=len(set(cols.tolist())),
                   row_min=int(rows.min()), row_max=int(rows.max()), rows_mod256=sorted(set((rows % 256).tolist()))[:12],
                   sample=[(int(r), int(c), float(out.view(-1, N)[r, c]), float(v.view(-1, N)[r, c])) for r, c in zip(rows[:6].tolist(), cols[:6].tolist())])
    print(json.dumps(rec), flush=True)
