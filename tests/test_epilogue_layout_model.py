"""Executable model of the lane exchanges of gemm256x's bf16 epilogue (open_sora_amd/csrc/gemm_epilogue16.h): the accumulator
layout of v_mfma_f32_16x16x32_bf16 with swapped operands -> stores of 8 rows x 128 bytes (round 4), and round 5's residual path
that LOADS in that store layout and walks the same exchanges backwards (both are involutions).  Pure numpy, no GPU: it pins the
index arithmetic the kernel relies on -- which (row, column) every register of every lane holds at every stage -- so an edit of the
epilogue can be checked here before its first GPU run (the approach of tests/conv_sw_emulator.py).  The GPU-side proof is
tests/test_gpu_kernels.py -k gemm (in-place residual GEMMs against the fp64 reference)."""
import numpy as np

NB = 8          # 16-blocks per wave-tile side (GeoX::NB): the wave tile is 128 x 128
LANES = 64


def lane_ids():
    lane = np.arange(LANES)
    return lane, lane >> 4, lane & 15      # lane, q4, l15


def permlane16_swap(x, y):
    """v_permlane16_swap vdst = x, src = y over the four 16-lane rows: x' = [x.r0, y.r0, x.r2, y.r2], y' = [x.r1, y.r1, x.r3, y.r3]"""
    x, y = x.reshape(4, 16, -1), y.reshape(4, 16, -1)
    xo = np.stack([x[0], y[0], x[2], y[2]]).reshape(LANES, -1)
    yo = np.stack([x[1], y[1], x[3], y[3]]).reshape(LANES, -1)
    return xo, yo


def quad_transpose(regs):
    """regs[r][lane]: X[r] of lane j -> X[j] of lane r inside every quad of lanes (osk_common.h::quad_transpose)"""
    out = [np.empty_like(regs[0]) for _ in range(4)]
    for lane in range(LANES):
        a, j = lane & ~3, lane & 3
        for r in range(4):
            out[r][lane] = regs[j][a + r]
    return out


def acc_tile(C, J, I):
    """accumulator layout: of tile (J, I) a lane owns row 16 I + l15 and the 4 columns 16 J + 4 q4 .. + 3 -> [lane, 4]"""
    _, q4, l15 = lane_ids()
    return np.stack([C[16 * I + l15, 16 * J + 4 * q4 + i] for i in range(4)], 1)


def chunk_interior(C, J, I):
    """pair_interior / chunk_interior: the two tiles (J, I), (J, I + 1) -> one 16-byte chunk per lane = 8 values [lane, 8]
    (x, y, z, w dwords = value pairs): sx = swap(a0[0:2], a1[0:2]), sy = swap(a0[2:4], a1[2:4]); chunk = (sx0, sy0, sx1, sy1)"""
    a0, a1 = acc_tile(C, J, I), acc_tile(C, J, I + 1)
    sx0, sx1 = permlane16_swap(a0[:, 0:2], a1[:, 0:2])
    sy0, sy1 = permlane16_swap(a0[:, 2:4], a1[:, 2:4])
    return np.concatenate([sx0, sy0, sx1, sy1], 1)


def test_chunk_is_eight_contiguous_columns_of_one_row():
    """the comment the store path relies on: lane (q4, l15)'s chunk of (J, I) = columns 16 J + 8 (q4 >> 1) .. + 7 of row 16 (I + (q4 & 1)) + l15"""
    C = np.arange(128 * 128, dtype=np.int64).reshape(128, 128)
    _, q4, l15 = lane_ids()
    for J in range(NB):
        for I in range(0, NB, 2):
            ch = chunk_interior(C, J, I)
            row = 16 * (I + (q4 & 1)) + l15
            col0 = 16 * J + 8 * (q4 >> 1)
            want = np.stack([C[row, col0 + i] for i in range(8)], 1)
            assert np.array_equal(ch, want), (J, I)


def store_addresses(I, crs):
    """row_pair_wide: after the quad transposes register 4 b + r of lane `lane` goes to element offset
    own + 16 j - j crs + r crs + 64 b, own = (16 (I + (q4 & 1)) + l15) crs + 8 (q4 >> 1), j = lane & 3 -> [lane, NB] offsets of the chunk's first element"""
    lane, q4, l15 = lane_ids()
    own = (16 * (I + (q4 & 1)) + l15) * crs + 8 * (q4 >> 1)
    j = lane & 3
    off = np.empty((LANES, NB), dtype=np.int64)
    for b in range(NB // 4):
        for r in range(4):
            off[:, 4 * b + r] = own + 16 * j - j * crs + r * crs + 64 * b
    return off


def test_wide_stores_cover_the_row_pair_once_in_128_byte_pieces():
    crs = 128                       # a wave tile as its own little matrix
    C = np.arange(128 * 128, dtype=np.int64).reshape(128, 128)
    for I in range(0, NB, 2):
        d = [chunk_interior(C, J, I) for J in range(NB)]                       # d[J][lane, 8]
        for b in range(NB // 4):
            for comp in range(4):                                              # the four dwords of the chunks, transposed separately
                regs = quad_transpose([d[4 * b + r][:, 2 * comp:2 * comp + 2] for r in range(4)])
                for r in range(4):
                    d[4 * b + r] = d[4 * b + r].copy()
                    d[4 * b + r][:, 2 * comp:2 * comp + 2] = regs[r]
        off = store_addresses(I, crs)
        out = np.full(128 * 128, -1, dtype=np.int64)
        for reg in range(NB):
            for lane in range(LANES):
                o = off[lane, reg]
                assert (out[o:o + 8] == -1).all(), "a store overlaps another"
                out[o:o + 8] = d[reg][lane]
            # one store instruction (all lanes, one register) = 8 rows x 128 contiguous bytes (64 bf16)
            rows = sorted(set(off[:, reg] // crs))
            assert len(rows) == 8
            for rw in rows:
                cols = np.sort(np.concatenate([np.arange(o % crs, o % crs + 8) for o in off[:, reg] if o // crs == rw]))
                assert len(cols) == 64 and cols[0] % 64 == 0 and np.array_equal(cols, np.arange(cols[0], cols[0] + 64))
        out = out.reshape(128, 128)
        lo, hi = 16 * I, 16 * I + 32
        assert np.array_equal(out[lo:hi], C[lo:hi]), I
        assert (out[:lo] == -1).all() and (out[hi:] == -1).all()


def test_residual_loaded_in_the_store_layout_comes_back_in_the_accumulator_layout():
    """round 5: rw[reg] = 16 bytes at the store address of register reg; inverse quad transposes; swap(rw.x, rw.z), swap(rw.y, rw.w)
    -> the 8-byte residual pieces of tiles (J, I) and (J, I + 1) in the accumulator layout (chunk_interior_res)"""
    crs = 128
    R = (np.arange(128 * 128, dtype=np.int64) * 7 + 3).reshape(128, 128)       # the residual
    flat = R.reshape(-1)
    for I in range(0, NB, 2):
        off = store_addresses(I, crs)
        rw = [np.stack([flat[off[:, reg] + i] for i in range(8)], 1) for reg in range(NB)]
        for b in range(NB // 4):
            for comp in range(4):
                regs = quad_transpose([rw[4 * b + r][:, 2 * comp:2 * comp + 2] for r in range(4)])
                for r in range(4):
                    rw[4 * b + r] = rw[4 * b + r].copy()
                    rw[4 * b + r][:, 2 * comp:2 * comp + 2] = regs[r]
        for J in range(NB):
            x, y, z, w = (rw[J][:, 2 * c:2 * c + 2] for c in range(4))
            ux0, ux1 = permlane16_swap(x, z)
            uy0, uy1 = permlane16_swap(y, w)
            assert np.array_equal(np.concatenate([ux0, uy0], 1), acc_tile(R, J, I)), (J, I)
            assert np.array_equal(np.concatenate([ux1, uy1], 1), acc_tile(R, J, I + 1)), (J, I)


# ---- the 32 x 32 accumulator layout of gemm256p / the fp8 kernel (open_sora_amd/csrc/gemm_epilogue.h): a lane owns ONE output row
# (l31 = lane % 32) and, per 8-column block qd = 0..3, the 4 columns qd 8 + hi 4 .. + 3 (hi = lane / 32).  The bf16 store pairs the
# blocks (qd, qd + 1) with v_permlane32_swap: the lower half-wave ends up with the whole block qd, the upper with block qd + 1.
def permlane32_swap(x, y):
    """v_permlane32_swap vdst = x, src = y: x' = [x.lo, y.lo], y' = [x.hi, y.hi] (halves of the wave)"""
    x, y = x.reshape(2, 32, -1), y.reshape(2, 32, -1)
    return np.stack([x[0], y[0]]).reshape(LANES, -1), np.stack([x[1], y[1]]).reshape(LANES, -1)


def acc32_block(C, qd):
    lane = np.arange(LANES)
    l31, hi = lane & 31, lane >> 5
    return np.stack([C[l31, qd * 8 + hi * 4 + i] for i in range(4)], 1)       # [lane, 4] = dwords (x | y) of packed[qd]


def test_32x32_layout_store_chunk_and_its_inverse_for_the_residual():
    C = np.arange(32 * 32, dtype=np.int64).reshape(32, 32)
    lane = np.arange(LANES)
    l31, hi = lane & 31, lane >> 5
    for qd in (0, 2):
        p0, p1 = acc32_block(C, qd), acc32_block(C, qd + 1)
        sx0, sx1 = permlane32_swap(p0[:, 0:2], p1[:, 0:2])
        sy0, sy1 = permlane32_swap(p0[:, 2:4], p1[:, 2:4])
        chunk = np.concatenate([sx0, sy0, sx1, sy1], 1)
        # stored at crow + (qd + hi) * 8: the 8 contiguous columns of block qd + hi of the lane's row
        want = np.stack([C[l31, (qd + hi) * 8 + i] for i in range(8)], 1)
        assert np.array_equal(chunk, want), qd
        # residual: the 16 bytes at the same address, swapped back pairwise (x, z) and (y, w) = the 8-byte pieces of blocks qd, qd + 1
        x, y, z, w = (want[:, 2 * c:2 * c + 2] for c in range(4))
        ux0, ux1 = permlane32_swap(x, z)
        uy0, uy1 = permlane32_swap(y, w)
        assert np.array_equal(np.concatenate([ux0, uy0], 1), p0)
        assert np.array_equal(np.concatenate([ux1, uy1], 1), p1)
