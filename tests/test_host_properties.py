"""Property tests (hypothesis) of the host-side arithmetic that sits between the reference's API and the kernels: the product
functions (open_sora_amd/sampling.py, _C.conv_out_dims, seqpar.shard_range) against the oracle's restatements / the defining
formulas over random arguments.  CPU only."""
import torch
from hypothesis import given, settings, strategies as st

from open_sora_amd import sampling
from oracle import sampling_oracle as S


@settings(max_examples=60, deadline=None)
@given(n=st.integers(1, 60), hp=st.integers(1, 64), wp=st.integers(1, 64), T=st.integers(1, 40),
       alpha=st.one_of(st.none(), st.floats(0.5, 8.0)))
def test_schedule_equals_the_oracle_and_descends_from_one_to_zero(n, hp, wp, T, alpha):
    ts = sampling.get_schedule(n, hp * wp, T, shift_alpha=alpha)
    assert ts == S.schedule(n, hp * wp, T, shift_alpha=alpha)
    # the reference's own f32 arithmetic: a t / (1 + (a - 1) t) at t = 1 can land one ulp above 1
    assert len(ts) == n + 1 and abs(ts[0] - 1.0) <= 2e-7 and ts[-1] == 0.0
    assert all(a > b for a, b in zip(ts[:-1], ts[1:]))


@settings(max_examples=40, deadline=None)
@given(b=st.integers(1, 3), T=st.integers(1, 5), hp=st.integers(1, 6), wp=st.integers(1, 6), seed=st.integers(0, 10))
def test_pack_unpack_round_trip_and_oracle(b, T, hp, wp, seed):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(b, 16, T, 2 * hp, 2 * wp, generator=g)
    p = sampling.pack(z)
    assert p.shape == (b, T * hp * wp, 64) and torch.equal(p, S.pack(z))
    # unpack takes PIXEL sizes and the VAE compression of the reference (sampling.py:70-77): height = 16 hp
    assert torch.equal(sampling.unpack(p, 16 * hp, 16 * wp, T), z)


@settings(max_examples=80, deadline=None)
@given(T=st.integers(1, 40), H=st.integers(1, 70), W=st.integers(1, 70), st_=st.sampled_from([1, 2]), sh=st.sampled_from([1, 2]),
       up_t=st.booleans(), up_hw=st.booleans())
def test_conv_out_dims_is_pad_then_unpadded_conv(T, H, W, st_, sh, up_t, up_hw):
    """CausalConv3d pads (k-1, 0) in time and k/2 on each spatial side (replicate) and convolves unpadded: the output extent is
    (n - 1) // stride + 1 of the (virtually upsampled) input; the nearest upsample keeps frame 0 single (unet_causal_3d_blocks.py:136-150)"""
    from tests import cpu_ops

    Tu = 1 + 2 * (T - 1) if up_t else T
    Hu, Wu = (2 * H, 2 * W) if up_hw else (H, W)
    want = ((Tu - 1) // st_ + 1, (Hu - 1) // sh + 1, (Wu - 1) // sh + 1)
    assert tuple(cpu_ops.conv_out_dims(T, H, W, (st_, sh, sh), (up_t, up_hw))) == want


@settings(max_examples=80, deadline=None)
@given(P=st.sampled_from([1, 2, 4, 8]), per=st.integers(1, 50), txt=st.integers(0, 60))
def test_sequence_shards_tile_the_token_axis(P, per, txt):
    """seqpar.SeqPar.shard_range (distributed.py:604-619): the ranks' ranges tile [0, L) in rank order, or every rank reports None
    (one rank, or a rank that would hold no image token: L / P <= L_txt) -- the same decision on all ranks"""
    from open_sora_amd import seqpar

    L = P * per
    ranges = []
    for r in range(P):
        sp = seqpar.SeqPar.__new__(seqpar.SeqPar)
        sp.P, sp.rank = P, r
        ranges.append(sp.shard_range(L, txt))
    if P == 1 or per <= txt:
        assert all(x is None for x in ranges)
    else:
        assert [x[0] for x in ranges] == [r * per for r in range(P)] and [x[1] for x in ranges] == [(r + 1) * per for r in range(P)]
        assert all(hi - max(lo, txt) > 0 for lo, hi in ranges)       # every rank keeps image tokens
