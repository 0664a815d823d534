"""oracle/sampling_oracle.py (the CPU restatement of the sampling loop that the GPU tests compare against) pinned to
the reference's OWN source text: I2VDenoiser.denoise, get_schedule, pack, unpack of opensora/utils/sampling.py are
executed from /root/reference through oracle.ref_loader.extract_defs.  CPU only; skipped where the reference is not
mounted (the GPU box)."""
import pytest
import torch

from oracle import ref_loader, sampling_oracle as S

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not mounted")
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ref():
    from tests.test_reference_api_dropin import _ref_namespace

    return _ref_namespace()


def test_schedule_pack_unpack_match_the_reference(ref):
    for n, L, T in ((30, 1024, 16), (3, 24, 3), (50, 3600, 33)):
        assert S.schedule(n, L, T) == ref["get_schedule"](n, L, T)
    assert S.schedule(7, 100, 5, shift=False) == ref["get_schedule"](7, 100, 5, shift=False)
    assert S.schedule(7, 100, 5, shift_alpha=2.5) == ref["get_schedule"](7, 100, 5, shift_alpha=2.5)
    z = torch.randn(2, 16, 3, 8, 12)
    assert torch.equal(S.pack(z), ref["pack"](z))
    assert torch.equal(S.unpack(S.pack(z), 4, 6, 3), z)
    assert torch.equal(ref["unpack"](S.pack(z), 8 * 16 // 2, 12 * 16 // 2, 3), z)


@pytest.mark.parametrize("dtype", [torch.float32, BF])
@pytest.mark.parametrize("osci", [False, True])
def test_i2v_denoise_is_the_reference_loop_bit_for_bit(ref, dtype, osci):
    """same stub model, same inputs: the restated loop equals I2VDenoiser.denoise exactly, in fp32 and in bf16
    (same operations in the same order and dtype)."""
    torch.manual_seed(0)
    n, T, Hh, Ww = 2, 3, 4, 6
    z = torch.randn(n, 16, T, Hh, Ww).to(dtype)
    img = ref["pack"](z).repeat(3, 1, 1)
    masks = torch.zeros(n, 1, T, Hh, Ww, dtype=dtype)
    masks[:, :, 0] = 1
    masked_ref = (torch.randn(n, 16, T, Hh, Ww) * masks.float()).to(dtype)
    ts = ref["get_schedule"](5, (Hh // 2) * (Ww // 2), T)
    W = torch.randn(64 + 68, 64) * 0.02

    def model(img, cond, timesteps, guidance, **kw):
        return (torch.cat([img.float(), cond.float()], -1) @ W * (1 + timesteps.float()[:, None, None])).to(img.dtype)

    kw = dict(text_osci=osci, image_osci=osci, scale_temporal_osci=osci)
    want = ref["I2VDenoiser"]().denoise(model, img=img, timesteps=ts, guidance=7.5, guidance_img=3.0, masks=masks,
                                        masked_ref=masked_ref, sigma_min=1e-5, **kw)
    got = S.i2v_denoise(model, img, ts, 7.5, 3.0, masks, masked_ref, **kw)
    assert got.dtype == want.dtype and torch.equal(got, want)
