"""CPU: host logic of the opt-in fp8 mode (MMDiTModel.enable_fp8) through the CPU emulation of the kernels'
semantics (tests/cpu_ops.py): which Linear layers are routed to the fp8 GEMM, the bf16 route for the shapes that
kernel refuses, weight row slices under the sequence-parallel call order, and the size of the quantisation error
against the bf16 mode and the fp32 oracle.  The kernels themselves: tests/test_gpu_fp8.py."""
import pytest
import torch

from oracle import configs, mmdit_oracle as O
from tests import cpu_ops
from tests.util import torch_inputs, torch_params

BF = torch.bfloat16
CFG = dict(configs.GOLDEN["hd128_eager_fused"][0], depth=1, depth_single_blocks=1)
GEOM = (2, 2, 12, 12, 160)      # B, T, h, w, L_txt: 576 image rows and 320 text rows per GEMM (>= 256)


class _Counting:
    """cpu_ops with call counters on the two GEMM entry points"""

    def __init__(self):
        self.n_fp8 = self.n_bf16 = 0
        self.bf16_shapes = []

    def __getattr__(self, name):
        return getattr(cpu_ops, name)

    def gemm(self, a, w, bias, out, **kw):
        self.n_bf16 += 1
        self.bf16_shapes.append((a.shape[0] * a.shape[1], w.shape[0], a.shape[2]))
        return cpu_ops.gemm(a, w, bias, out, **kw)

    def gemm_pair(self, first, second, *, gelu_from=None):     # bf16 mode: the img / txt Linear pairs of a double block
        for d in (first, second):
            self.gemm(d["a"], d["w"], d["bias"], d["out"], gelu_from=gelu_from,
                      **{k: d[k] for k in ("res", "gate", "gate_batch_stride") if k in d})

    def gemm_group(self, tasks):     # bf16 mode (round 6): a block's projection as a task list -- the V^T task belongs to the same Linear
        ok = cpu_ops.gemm_group(tasks)      # as the plain task of its stream: one count per Linear layer, as before
        if ok:
            for d in tasks:
                if "vt" not in d:
                    self.n_bf16 += 1
                    self.bf16_shapes.append((d["a"].shape[0] * d["a"].shape[1], d["w"].shape[0], d["a"].shape[2]))
        return ok

    def gemm_fp8(self, *a, **kw):
        self.n_fp8 += 1
        return cpu_ops.gemm_fp8(*a, **kw)

    def attention_fwd_pv8(self, *a, **kw):
        self.n_pv8 = getattr(self, "n_pv8", 0) + 1
        return cpu_ops.attention_fwd_pv8(*a, **kw)


@pytest.fixture()
def counting_mmdit(hip_lib):
    from open_sora_amd import mmdit

    ops = _Counting()
    mmdit.set_ops_for_testing(ops)
    yield mmdit, ops
    mmdit.set_ops_for_testing(hip_lib)


def _model(mmdit, cfg=CFG):
    model = mmdit.Flux(device_map="cpu", torch_dtype=BF, **cfg)
    model.load_state_dict(torch_params(cfg, dtype=BF), strict=True)
    return model


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def test_fp8_mode_routes_block_linears_and_bounds_the_error(counting_mmdit):
    mmdit, ops = counting_mmdit
    model = _model(mmdit)
    inp = torch_inputs(CFG, *GEOM, dtype=BF)
    with torch.inference_mode():
        ref16 = model(**inp).clone()
        n16 = ops.n_bf16
        assert ops.n_fp8 == 0
        model.enable_fp8()
        ops.n_bf16 = 0
        ops.bf16_shapes.clear()
        out8 = model(**inp).clone()
        truth = O.forward(torch_params(CFG), CFG, **torch_inputs(CFG, *GEOM))
    # double block: 2 streams x (qkv, proj, mlp up, mlp down); single block: linear1, linear2
    assert ops.n_fp8 == 2 * 4 + 2
    assert ops.n_bf16 == n16 - ops.n_fp8          # embedders and the final layer stay bf16
    assert ops.n_pv8 == 2                         # both blocks' attention with the fp8 P.V product (head_dim 128)
    e8, e16 = _rel(out8, truth), _rel(ref16, truth)
    assert e16 <= 2e-2 and e8 <= 5e-2, (e8, e16)  # fp8 gate = relL2 <= 5e-2
    assert _rel(out8, ref16) <= 5e-2
    model.enable_fp8(False)
    ops.n_fp8 = 0
    with torch.inference_mode():
        again = model(**inp)
    assert ops.n_fp8 == 0 and torch.equal(again, ref16)


def test_fp8_mode_small_shapes_stay_on_the_bf16_gemm(counting_mmdit):
    mmdit, ops = counting_mmdit
    model = _model(mmdit).enable_fp8()
    small = (1, 1, 6, 6, 24)         # 36 image rows, 24 text rows: below the fp8 kernel's 256-row tile
    with torch.inference_mode():
        out = model(**torch_inputs(CFG, *small, dtype=BF))
        ref = O.forward(torch_params(CFG, dtype=BF), CFG, **torch_inputs(CFG, *small, dtype=BF))
    assert ops.n_fp8 == 0 and torch.isfinite(out.float()).all()
    assert _rel(out, ref) <= 2e-2


def test_fp8_weight_row_slices_are_views(counting_mmdit):
    mmdit, _ = counting_mmdit
    w = torch.randn(512, 256).to(BF)
    fw = mmdit.Fp8Weight(w)
    part = fw[256:]
    assert part.shape == (256, 256)
    assert part.w8.data_ptr() == fw.w8[256:].data_ptr() and part.sw.data_ptr() == fw.sw[256:].data_ptr()
    deq = part.w8.view(torch.float8_e4m3fn).float() * part.sw[:, None]
    assert _rel(deq, w[256:].float()) <= 4e-2
