"""The LDS image of the large-tile GEMM / conv kernels (rows of 128 bytes = 64 bf16 k, 16-byte chunk c of row r stored at
chunk position c ^ ((r >> 1) & 7)) must be conflict-free for BOTH fragment lane maps that read it with ds_read_b128:
  * v_mfma_f32_32x32x16_bf16 (gemm256 [fp8] / gemm256p; rounds 1-2 also: gemm256w, conv256t / conv256w, since removed): lane l reads row l % 32, chunk 2 ks + l / 32;
  * v_mfma_f32_16x16x32_bf16 (gemm256x, conv256x):                               lane l reads row l % 16, chunk 4 s + l / 16.
ds_read_b128 is served in four groups of 16 lanes (/opt/skills/guides/MI355X_MICROARCH.md, LDS section); lanes of one group
conflict when they hit the same 16-byte slot of the 256-byte bank window with different addresses.  Pure arithmetic: runs on CPU."""
import pytest

GROUPS = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
    list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
    list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64)),
]


def image_addr(row: int, chunk: int) -> int:
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)


def conflicts(addr_of_lane) -> int:
    extra = 0
    for g in GROUPS:
        slots = {}
        for lane in g:
            a = addr_of_lane(lane)
            slots.setdefault((a >> 4) & 15, set()).add(a)
        extra += sum(len(v) - 1 for v in slots.values())
    return extra


def test_lane_groups_partition_the_wave():
    assert sorted(sum(GROUPS, [])) == list(range(64)) and all(len(g) == 16 for g in GROUPS)


@pytest.mark.parametrize("ks", range(4))
@pytest.mark.parametrize("base_row", [0, 32, 96, 224])
def test_32x32x16_fragment_reads_are_conflict_free(ks, base_row):
    assert conflicts(lambda l: image_addr(base_row + (l & 31), 2 * ks + (l >> 5))) == 0


@pytest.mark.parametrize("s", range(2))
@pytest.mark.parametrize("base_row", [0, 16, 112, 240])
def test_16x16x32_fragment_reads_are_conflict_free(s, base_row):
    assert conflicts(lambda l: image_addr(base_row + (l & 15), 4 * s + (l >> 4))) == 0


def test_second_substep_address_is_the_first_one_xor_64():
    """gen_x4 derives the k 32..63 fragment address as faA0 ^ 64 (chunk + 4 == chunk ^ 4 for chunk < 4, and the swizzle is an XOR)"""
    for row in range(256):
        for q4 in range(4):
            assert image_addr(row, 4 + q4) == image_addr(row, q4) ^ 64


def test_unswizzled_image_would_conflict():
    """the check has teeth: without the XOR the 16-row map is 8-way conflicted"""
    assert conflicts(lambda l: (l & 15) * 128 + ((l >> 4) << 4)) > 0


def test_lds_dma_piece_writes_whole_swizzled_rows():
    """an LDS-DMA instruction of the loaders covers 8 tile rows: lane l writes row l / 8, position l % 8, and fetches the chunk
    that belongs there (position ^ key): every (row, chunk) of the 8 rows exactly once"""
    for j in range(32):                       # instruction j covers rows 8 j .. 8 j + 7
        seen = set()
        for l in range(64):
            r = 8 * j + (l >> 3)
            c = (l & 7) ^ ((r >> 1) & 7)      # source chunk of this lane (csrc/gemm256*.hip: `spos ^ ((r >> 1) & 7)`)
            assert image_addr(r, c) == r * 128 + ((l & 7) << 4)   # lands at the lane's linear LDS position
            seen.add((r, c))
        assert len(seen) == 64


def test_epilogue_quad_transpose_store_map():
    """gemm_epilogue16.h::row_pair_wide / osk_common.h::quad_transpose, restated: after the lane-row exchange lane (q4, l15) holds,
    for the row-block pair (I, I + 1) and column block J, the 16-byte chunk (row 16 (I + (q4 & 1)) + l15, columns 16 J + 8 (q4 >> 1));
    two DPP butterfly steps (quad_perm [1,0,3,2] against register bit 0, [2,3,0,1] against bit 1) transpose (register, lane-of-quad);
    the store of register r of group b then goes to (own row - j + r, columns 64 b + 16 j + 8 (q4 >> 1)), j = lane % 4.  Checked: the
    label that arrives in every register IS the address it is stored to, the 128 x 128 wave tile is covered exactly once, and one
    store instruction touches 8 rows x 128 contiguous bytes (round 3: 32 rows x 32 bytes)."""
    import numpy as np

    lanes = np.arange(64)
    q4, l15, j = lanes >> 4, lanes & 15, lanes & 3
    odd, hi = (lanes & 1).astype(bool), (lanes & 2).astype(bool)
    swap1, swap2 = (lambda v: v[lanes ^ 1]), (lambda v: v[lanes ^ 2])
    seen = set()
    for I in (0, 2, 4, 6):
        own = 16 * (I + (q4 & 1)) + l15
        d = [own * 1000 + 16 * J + 8 * (q4 >> 1) for J in range(8)]            # label = row * 1000 + first column
        for b in range(2):
            x0, x1, x2, x3 = d[4 * b: 4 * b + 4]
            s0, s1, s2, s3 = swap1(x0), swap1(x1), swap1(x2), swap1(x3)
            y0, y1, y2, y3 = np.where(odd, s1, x0), np.where(odd, x1, s0), np.where(odd, s3, x2), np.where(odd, x3, s2)
            t0, t1, t2, t3 = swap2(y0), swap2(y1), swap2(y2), swap2(y3)
            d[4 * b: 4 * b + 4] = [np.where(hi, t2, y0), np.where(hi, t3, y1), np.where(hi, y2, t0), np.where(hi, y3, t1)]
        for b in range(2):
            for r in range(4):
                row, col = own - j + r, 64 * b + 16 * j + 8 * (q4 >> 1)
                assert (d[4 * b + r] == row * 1000 + col).all(), (I, b, r)
                pieces = {}
                for ln in lanes:
                    key = (int(row[ln]), int(col[ln]))
                    assert key not in seen
                    seen.add(key)
                    pieces.setdefault(key[0], []).append(key[1])
                assert len(pieces) == 8 and all(sorted(c) == list(range(64 * b, 64 * b + 64, 8)) for c in pieces.values())
    assert len(seen) == 128 * 16
