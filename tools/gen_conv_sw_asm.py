#!/usr/bin/env python
"""Generator of the hand-scheduled gfx950 K loop of the SLIDING-WINDOW CausalConv3d (open_sora_amd/csrc/conv3d_256.hip,
convsw_kernel): 3 x 3 x 3, stride 1 (plain; the fused-upsample form `up` reads a 10 x 10 source patch), a workgroup tile = the 16 x 16 spatial brick of ONE output frame x
256 (NBJ = 8) or 128 (NBJ = 4) output channels.

Why: conv256x_kernel (tools/gen_gemm_asm.py::gen_conv_x4) is an implicit GEMM whose every filter tap re-fetches its 256 x 64
activation tile through L2 -- 27 fetches of (almost) the same voxels, fabric-side 103 GB per VAE encode + decode against ~2 GB
of activations, 57 % of the wave cycles waiting (profiles/r02_pmc_conv.txt).  Here the activations of a tile cross the fabric
ONCE per 32-channel block: the 3-frame x 18 x 18 halo brick of the block sits in LDS (64 bytes per voxel) and the 27 taps
differ only in the IMMEDIATE offset of their fragment reads.

K axis = channel block (32 channels, outer) x tap (inner, frame-major: tap = 9 dt + 3 dh + dw).  One step = one tap = one
v_mfma_f32_16x16x32_bf16 k-step over the wave tile (128 voxels x 16 NBJ channels: 8 x NBJ MFMAs, accumulators in AGPRs, the
same accumulator layout as conv256x_kernel, so the epilogue is shared).

LDS (all filled by LDS-DMA, global_load_lds_dwordx4, 1 KiB per wave instruction):
  halo   3 frame slots x 21 KiB: voxel v = 18 hh + ww of slot dt at (336 dt + v) * 64, 16-byte chunk c of a voxel at position
         c ^ ((ww >> 1) & 3) (conflict-free for the 16-row x 32-k lane map at every tap shift, tests/test_conv_sw_model.py);
         ROLLING refill: slot dt is last read by tap 9 dt + 8, so the NEXT channel block's frame dt streams in during taps
         9 dt + 9 .. + 2 of this block -- one halo buffer, 17 taps of slack.
  W ring 6 stages x (32 NBJ rows x 64 bytes): stage s % 6 holds the weights of step s, fetched 5 steps ahead (the stage is free
         as soon as the barrier of step s has passed: its last reader was step s - 1).
The body is TWO channel blocks = 54 steps, fully unrolled (immediate offsets, stage numbers, fragment-set parity and the vmcnt
of every step are static); Cin / 32 is even for every supported layer.

One step of a wave:   s_waitcnt vmcnt(N) ; s_barrier          -- the weights of step s + 1 have landed for every wave
  MFMA groups j = 0 .. NBJ - 1 of 8 (one weight fragment against the 8 activation fragments); in their shadows:
    the 8 activation fragment reads of step s + 1 (other fragment set), weight fragment j of step s + 1 into the register quad
    group j has just released (one set of weight fragments, rolling), the LDS-DMA pieces of step s + 5's weights and -- on 9 of
    27 steps -- two halo pieces, the scalar address advances.
GN form (`gn`, round 4; plain NBJ = 8 and two-frame geometry): the conv's INPUT is silu(GroupNorm(x)) and the kernel reads x itself --
the normalise + SiLU pass (gn_apply_kernel: 4 bytes of HBM per element and norm) is folded into the halo refill.  The halo pieces are
fetched into a ring of VGPR quads (global_load_dwordx4, same voxel order, a lane always holds chunk lane % 4 = 8 fixed channels of
the 32-channel block), transformed LAT steps later in the MFMA shadows with the reference's rounding points
(bf16(x a_c + d_c) -> y sigmoid(y) -> bf16; a, d from a per-(batch, channel) table, 16 VGPRs reloaded per channel block) and
written with ds_write_b128 to the swizzled position the LDS-DMA form fills (the swizzle moves from the global offset to the LDS
address); the write is published by the barrier two steps later.  72 VALU (16 transcendental) per piece: ~1 per MFMA on average.
The generator keeps its own model of what every instruction does (`Op.meta`); tests/test_conv_sw_model.py executes that model
symbolically (every accumulator tile receives every (channel block, tap) product exactly once, no fragment is used before its
wait, no LDS region is refilled before a barrier behind its last read or read before its fill is published).
"""
import argparse
import os

NB = 8                      # 16-voxel row blocks per wave tile
TAPS = 27
BODY = 2 * TAPS             # steps per loop body (two channel blocks)
# body step at which the refill of a halo frame slot starts (three steps, two pieces per wave each): right behind the first
# barrier after the slot's last reader (tap 9 slot + 8 of either block) has retired its fragment reads
REFILL_START = {1: {0: 2, 9: 0, 18: 1, 27: 2, 36: 0, 45: 1}, 2: {0: 2, 10: 0, 18: 1, 28: 2, 36: 0, 46: 1}}
# two-frame form (4 frame slots; the second frame's waves read slot dt + 1 where the first frame's read slot dt)
REFILL_START_F2 = {0: 2, 3: 3, 9: 0, 18: 1, 27: 2, 30: 3, 36: 0, 45: 1}

V_OPERANDS = ["xa0", "xa1", "xa2", "yb", "woff0", "woff1", "woff2", "woff3", "hoff0", "hoff1", "hoff2", "hoff3", "hoff4", "hoff5"]
S_OPERANDS = ["wbase", "xb0", "xb1", "xb2", "xb3", "cin2", "nbody", "wdst", "hdst", "hdst5"]
OPERANDS = V_OPERANDS + S_OPERANDS
GN_EXP = {}                                      # experiments (tools only): {"windows": {f2: {start: (slot, T0, T1)}}, "valu": "none" | "notrans"}
# GN form: per-lane LDS write address of each halo piece, per-lane byte offset into a channel block's (scale, shift) rows; table base
GN_V_OPERANDS = ["hdw0", "hdw1", "hdw2", "hdw3", "hdw4", "hdw5", "goff"]
GN_S_OPERANDS = ["gbase"]

# asm-owned scalars
S_WB, S_X0, S_X1, S_X2 = 40, 42, 44, 46          # 64-bit running pointers: weights of the next step to fetch, halo frame bases
S_CIN2, S_NBODY, S_IT = 48, 49, 50
S_WDST, S_HDST, S_HDST5 = 51, 52, 53
S_WRAP, S_WRAPL = 54, 56                         # 64-bit: 64 - 26 cin2 (next block, tap 0) / -26 cin2 (same block, tap 0)
S_TMP, S_T2, S_T3 = 58, 59, 60
S_X3 = 62
S_G = 64                                         # GN form, 64-bit: the (scale, shift) rows of the channel block loaded last
S_MASK, S_NL2E = 66, 67                          # GN form: 0xffff0000, -log2(e) (4-byte encodings instead of 8 with a literal)
S_FIRST, S_LAST = 40, 63
S_X = [S_X0, S_X1, S_X2, S_X3]


def vr(b, n=1):
    return "v%d" % b if n == 1 else "v[%d:%d]" % (b, b + n - 1)


def ar(b, n=1):
    return "a%d" % b if n == 1 else "a[%d:%d]" % (b, b + n - 1)


class Cfg:
    def __init__(self, nbj, up=False, f2=False, gn=False):
        self.NBJ = nbj
        self.GN = gn                              # silu(GroupNorm(.)) of the input folded into the halo refill (module docstring)
        assert not gn or (nbj == 8 and not up)
        self.UP = up                              # the decoder's nearest 2x (H, W) upsample folded into the halo (see a_offset)
        # F2: the tile is the 16 x 16 brick of TWO consecutive output frames x 128 channels: waves (frame, brick half), each
        # 128 voxels x 128 channels; four frame slots; per-tile fixed costs and the weight traffic are shared by 512 voxels
        self.F2 = f2
        assert not f2 or (nbj == 8 and not up)
        self.NSLOT = 4 if f2 else 3
        self.BN = 128 if f2 else 32 * nbj         # output channels per workgroup tile
        # one halo frame slot: 18 x 18 voxels padded to 21 LDS-DMA pieces; upsampled: the 10 x 10 SOURCE voxels, 7 pieces
        self.PITCH, self.SLOTV, self.NPIECE = (10, 112, 7) if up else (18, 336, 21)
        self.SLOT = self.SLOTV * 64
        self.HALO = self.NSLOT * self.SLOT
        self.NT = 1 if up else 3                  # steps per slot refill (two pieces per wave and step)
        self.BAR = 1 if nbj == 8 else 2           # steps per barrier (64 MFMAs per wave between barriers in both forms)
        self.NS = 6 if nbj == 8 else 9            # W ring stages (one step each); BODY % NS == 0: stage numbers are static
        self.LEAD = self.NS - self.BAR            # step s fetches the weights of step s + LEAD into the stage of step s + LEAD - NS,
        assert BODY % self.NS == 0                # whose last reader retired before the latest barrier (<= s - s % BAR)
        self.HALO_STEPS = {}                      # body step -> (slot, third, the pointer advance behind it is conditional)
        starts = REFILL_START_F2 if f2 else REFILL_START[self.BAR]
        if gn:
            # GN form: the loads go to registers, so a refill group's fetch may start before the slot's last reader has retired
            # (only the LDS write must not); the groups are placed so that the transform work is EVENLY spread over the body --
            # a filler beside 16-cycle MFMAs is only hidden while a gap carries <= ~3 of them -- and the register ring never holds
            # more than NRING pieces.  GN_WINDOWS: block-relative first load step of a group -> (frame slot, [T0, T1) = the steps
            # (fractional: 64 MFMA gaps per step) over which its 6 pieces per wave are transformed and written, one after the other)
            self.GN_WINDOWS = ({0: (2, 2.0, 7.5), 5: (3, 8.0, 13.0), 10: (0, 14.0, 19.5), 16: (1, 19.5, 25.5)} if f2 else
                               {0: (2, 3.0, 10.5), 9: (0, 12.0, 19.5), 18: (1, 21.0, 26.5)})
            for k, v in GN_EXP.get("windows", {}).get(f2, {}).items():
                self.GN_WINDOWS[k] = v
            starts = {}
            for blk in (0, 1):
                for st, (sl, _, _) in self.GN_WINDOWS.items():
                    starts[st + blk * TAPS] = sl
        for start, slot in starts.items():
            for third in range(self.NT):
                self.HALO_STEPS[start + third] = (slot, third, (slot >= 2) == (start >= TAPS))
        self.W_STAGE = self.BN * 64
        self.W_BASE = self.HALO
        self.SMEM = self.HALO + self.NS * self.W_STAGE
        self.NWP = self.BN // 64                  # weight LDS-DMA pieces per wave and step
        self.NM = NB * nbj                        # MFMAs per step
        self.NACC = 4 * self.NM
        self.V0 = 96
        self.VA = [self.V0, self.V0 + 32]         # two activation fragment sets
        self.VB = self.V0 + 64                    # one rolling weight fragment set
        self.VY = self.VB + 4 * nbj               # weight fragment address of each ring stage (ds_read immediates are 16 bits)
        self.VN = 64 + 4 * nbj + self.NS
        self.tag = "sw%s%s%d" % ("g" if gn else "", "u" if up else "f" if f2 else "", self.BN)
        self.OPERANDS = V_OPERANDS + (GN_V_OPERANDS if gn else []) + S_OPERANDS + (GN_S_OPERANDS if gn else [])
        self.OPN = {n: "%%%d" % i for i, n in enumerate(self.OPERANDS)}
        self.S_LAST = S_NL2E if gn else S_LAST
        if gn:
            # the ring of register quads that holds the pieces between load and LDS write; pieces per body (36 / 48) % NRING == 0:
            # ring positions are static
            self.NRING = 8 if f2 else 6
            self.VST = self.V0 + self.VN
            self.VSC = self.VST + 4 * self.NRING  # 8 scales, 8 shifts of this lane's channels, 8 temporaries
            self.VSH, self.VT = self.VSC + 8, self.VSC + 16
            self.VN += 4 * self.NRING + 24
            assert self.V0 + self.VN <= 256
            # groups in body order: (first load step, slot, T0, T1, channel block it fetches: 0 = this body's first, 1 = its second,
            # 2 = the next body's first); the (scale, shift) rows change right behind the last transform of a block's groups
            label = lambda st, sl: (0 if sl >= 2 else 1) if st < TAPS else (1 if sl >= 2 else 2)
            self.GN_GROUPS = sorted((st + blk * TAPS, sl, t0 + blk * TAPS, t1 + blk * TAPS, label(st + blk * TAPS, sl))
                                    for blk in (0, 1) for st, (sl, t0, t1) in self.GN_WINDOWS.items())
            self.TAB_TIMES = {}                   # body time -> the advance is conditional (first block of the NEXT body)
            for g0, g1 in zip(self.GN_GROUPS, self.GN_GROUPS[1:]):
                assert g0[3] <= g1[2], "transform windows overlap"
                if g0[4] != g1[4]:
                    assert g1[4] == g0[4] + 1
                    self.TAB_TIMES[g0[3] + 0.05] = g1[4] == 2
            assert len(self.TAB_TIMES) == 2 and self.GN_GROUPS[-1][3] < BODY

    def a_offset(self, tap, i):
        """immediate of activation fragment i (brick rows 8 wm + i) of tap (dt, dh, dw), and the address operand it adds to.
        Plain: halo voxel (dt, i + dh, l15 + dw); the column shift dw is part of the immediate AND selects the operand (the swizzle
        key depends on the halo column).  Upsampled: output row i + dh - 1 of the brick reads SOURCE row (i + dh - 1) >> 1, i.e.
        halo row ((i + dh - 1) >> 1) + 1; the column map (l15 + dw - 1) >> 1 is per lane and lives in the operand."""
        dt, r9 = divmod(tap, 9)
        dh, dw = divmod(r9, 3)
        if self.UP:
            return (self.SLOTV * dt + (((i + dh - 1) >> 1) + 1) * self.PITCH) * 64, dw
        return (self.SLOTV * dt + (i + dh) * self.PITCH + dw) * 64, dw


class Op:
    __slots__ = ("text", "kind", "meta")

    def __init__(self, text, kind, meta=None):
        self.text, self.kind, self.meta = text, kind, meta


def generate(c):
    """-> list of Op.  kinds: M mfma, R ds_read, D lds-dma, m0, S salu, W waitcnt, B barrier, L label, J branch, X other"""
    ops = []
    OPN = c.OPN
    NS, LEAD, HALO_STEPS = c.NS, c.LEAD, c.HALO_STEPS
    pend = []          # LDS reads (GN form: and halo writes) in flight, in order (tags)
    vm = []            # LDS-DMA pieces (GN form: and register loads) in flight, in order (tags)
    ring = {}          # GN form: (load step, slot, piece) -> ring position of the register quad
    ring_pre = {}      # ... as the transform schedule assumed it
    nring = [0]

    def emit(text, kind="X", meta=None):
        ops.append(Op(text, kind, meta))

    def ds_read(dst, addr, imm, tag, meta):
        assert 0 <= imm < 65536, imm
        emit("ds_read_b128 %s, %s offset:%d" % (vr(dst, 4), addr, imm), "R", dict(meta, dst=dst, tag=tag))
        pend.append(tag)

    def need(tag):
        if tag in pend:
            idx = len(pend) - 1 - pend[::-1].index(tag)
            n = min(len(pend) - 1 - idx, 15)                 # (a 4-bit counter: the wait retires at least what is asked for)
            emit("s_waitcnt lgkmcnt(%d)" % n, "W", dict(lgkm=n))
            del pend[: len(pend) - n]

    def a_read(step, i):       # activation fragment i of step `step` (body-relative, may be BODY = next body's step 0)
        tap = step % TAPS
        off, dw = c.a_offset(tap, i)
        ds_read(c.VA[step % 2] + 4 * i, OPN["xa%d" % dw], off, ("A", step, i),
                dict(region=("H", tap // 9), slots=(tap // 9, tap // 9 + 1) if c.F2 else (tap // 9,), step=step, frag=("A", i), off=off, dw=dw))

    def b_read(step, j):
        ds_read(c.VB + 4 * j, vr(c.VY + step % NS), j * 1024, ("B", step, j),
                dict(region=("W", step % NS), step=step, frag=("B", j)))

    def mfma(step, j, i):
        acc = ar((j * NB + i) * 4, 4)
        emit("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (acc, vr(c.VB + 4 * j, 4), vr(c.VA[step % 2] + 4 * i, 4), acc), "M",
             dict(step=step, j=j, i=i, a=c.VA[step % 2] + 4 * i, b=c.VB + 4 * j))

    def w_pieces(fstep):       # LDS-DMA of the weights of step `fstep` (body-relative; the running pointer S_WB addresses them)
        out = []
        for k in range(c.NWP):
            out.append((Op("s_add_u32 m0, s%d, %d" % (S_WDST, c.W_BASE + (fstep % NS) * c.W_STAGE + k * 4096), "m0"),
                        Op("global_load_lds_dwordx4 %s, s[%d:%d]" % (OPN["woff%d" % k], S_WB, S_WB + 1), "D",
                           dict(region=("W", fstep % NS), fstep=fstep, tag=("W", fstep)))))
        return out

    def g_load(slot, k, key, pro=None):
        """GN form: halo piece k of frame slot `slot` into the next quad of the register ring (pro = n: the prologue's n-th piece, into
        the fragment registers v[V0 + 4 n ..] -- nothing reads fragments before the prologue's barrier)"""
        if pro is not None:
            return Op("global_load_dwordx4 %s, %s, s[%d:%d]" % (vr(c.V0 + 4 * pro, 4), OPN["hoff%d" % k], S_X[slot], S_X[slot] + 1), "G",
                      dict(region=("H", slot), tag=("G",) + key, piece=k, ring=("pro", pro), dst=c.V0 + 4 * pro))
        r = nring[0] % c.NRING
        nring[0] += 1
        ring[key] = r
        assert ring_pre.get(key, r) == r, (key, r, ring_pre[key])
        return Op("global_load_dwordx4 %s, %s, s[%d:%d]" % (vr(c.VST + 4 * r, 4), OPN["hoff%d" % k], S_X[slot], S_X[slot] + 1), "G",
                  dict(region=("H", slot), tag=("G",) + key, piece=k, ring=r, dst=c.VST + 4 * r))

    def xform(slot, key, r, d=None):
        """GN form: the piece in ring quad r -> list of Op: 72 VALU (two dwords at a time, interleaved instruction by instruction:
        a transcendental's result is never read by the next instruction -- the gfx940 forwarding hazard), then its ds_write_b128.
        y = bf16(x a + d); out = bf16(y / (1 + 2^(-y log2 e)))  (csrc/groupnorm.hip::gn_apply_kernel, osk_common.h::silu)"""
        out = []
        d = c.VST + 4 * r if d is None else d
        mode = GN_EXP.get("valu")          # experiments: "none" = loads and writes only, "notrans" = v_mov for v_exp / v_rcp

        def each(P, fmt, **flags):
            for w, t in P:
                text = fmt(w, t)
                if mode == "notrans" and text.startswith(("v_exp", "v_rcp")):
                    text = "v_mov_b32_e32 " + text.split(" ", 1)[1]
                meta = dict(key=key, quad=r)
                if flags.get("src"):
                    meta["src"] = w
                if flags.get("tab"):
                    meta["tab"] = True
                if flags.get("dst"):
                    meta["dstw"] = w
                out.append(Op(text, "V", meta))

        for w0 in (0, 2):
            P = [(w0, c.VT), (w0 + 1, c.VT + 4)]
            each(P, lambda w, t: "v_lshlrev_b32_e32 %s, 16, %s" % (vr(t), vr(d + w)), src=1)
            each(P, lambda w, t: "v_and_b32_e32 %s, s%d, %s" % (vr(t + 1), S_MASK, vr(d + w)), src=1)
            each(P, lambda w, t: "v_fma_f32 %s, %s, %s, %s" % (vr(t), vr(t), vr(c.VSC + 2 * w), vr(c.VSH + 2 * w)), tab=1)
            each(P, lambda w, t: "v_fma_f32 %s, %s, %s, %s" % (vr(t + 1), vr(t + 1), vr(c.VSC + 2 * w + 1), vr(c.VSH + 2 * w + 1)), tab=1)
            each(P, lambda w, t: "v_cvt_pk_bf16_f32 %s, %s, %s" % (vr(t), vr(t), vr(t + 1)))
            each(P, lambda w, t: "v_and_b32_e32 %s, s%d, %s" % (vr(t + 1), S_MASK, vr(t)))
            each(P, lambda w, t: "v_lshlrev_b32_e32 %s, 16, %s" % (vr(t), vr(t)))
            each(P, lambda w, t: "v_mul_f32_e32 %s, s%d, %s" % (vr(t + 2), S_NL2E, vr(t)))
            each(P, lambda w, t: "v_mul_f32_e32 %s, s%d, %s" % (vr(t + 3), S_NL2E, vr(t + 1)))
            each(P, lambda w, t: "v_exp_f32_e32 %s, %s" % (vr(t + 2), vr(t + 2)))
            each(P, lambda w, t: "v_exp_f32_e32 %s, %s" % (vr(t + 3), vr(t + 3)))
            each(P, lambda w, t: "v_add_f32_e32 %s, 1.0, %s" % (vr(t + 2), vr(t + 2)))
            each(P, lambda w, t: "v_add_f32_e32 %s, 1.0, %s" % (vr(t + 3), vr(t + 3)))
            each(P, lambda w, t: "v_rcp_f32_e32 %s, %s" % (vr(t + 2), vr(t + 2)))
            each(P, lambda w, t: "v_rcp_f32_e32 %s, %s" % (vr(t + 3), vr(t + 3)))
            each(P, lambda w, t: "v_mul_f32_e32 %s, %s, %s" % (vr(t), vr(t), vr(t + 2)))
            each(P, lambda w, t: "v_mul_f32_e32 %s, %s, %s" % (vr(t + 1), vr(t + 1), vr(t + 3)))
            each(P, lambda w, t: "v_cvt_pk_bf16_f32 %s, %s, %s" % (vr(d + w), vr(t), vr(t + 1)), dst=1)
        if mode == "none":
            out = []
        k = key[-1]
        out.append(Op("ds_write_b128 %s, %s offset:%d" % (OPN["hdw%d" % k], vr(d, 4), slot * c.SLOT), "Wd",
                      dict(region=("H", slot), tag=("HW",) + key, piece=k, src=d, key=key, quad=r)))
        return out

    def table_load(cond):
        """GN form: advance the table pointer by one channel block (256 bytes; `cond`: only if another body iteration follows) and
        fetch this lane's 8 scales + 8 shifts"""
        out = list(add64(S_G, "256")) if not cond else cond_next() + [Op("s_cselect_b32 s%d, 256, 0" % S_T2, "S")] + add64(S_G, "s%d" % S_T2)
        for q in range(4):
            out.append(Op("global_load_dwordx4 %s, %s, s[%d:%d]%s" % (vr(c.VSC + 4 * q, 4), OPN["goff"], S_G, S_G + 1,
                                                                     " offset:%d" % (16 * q) if q else ""), "G",
                          dict(table=True, tag=("T", q), dst=c.VSC + 4 * q)))
        return out

    def h_pieces(slot, third, tag):
        out = []
        for k in (2 * third, 2 * third + 1):      # piece k of wave w = 1-KiB block min(4 k + w, NPIECE - 1) of the slot
            base, imm = (S_HDST, slot * c.SLOT + k * 4096) if 4 * k + 3 < c.NPIECE else (S_HDST5, slot * c.SLOT)
            out.append((Op("s_add_u32 m0, s%d, %d" % (base, imm), "m0"),
                        Op("global_load_lds_dwordx4 %s, s[%d:%d]" % (OPN["hoff%d" % k], S_X[slot], S_X[slot] + 1), "D",
                           dict(region=("H", slot), tag=tag, piece=k))))
        return out

    def add64(reg, lo, hi=None):
        """reg(64) += (lo, hi) ; hi None = immediate / zero-extended 32-bit register"""
        out = [Op("s_add_u32 s%d, s%d, %s" % (reg, reg, lo), "S")]
        out.append(Op("s_addc_u32 s%d, s%d, %s" % (reg + 1, reg + 1, "0" if hi is None else hi), "S"))
        return out

    def cond_next():           # scc = (another body iteration follows)
        return [Op("s_add_u32 s%d, s%d, 1" % (S_TMP, S_IT), "S"), Op("s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NBODY), "S")]

    def w_advance(fstep):      # after the pieces of step fstep: point at step fstep + 1
        tap = fstep % TAPS
        if tap != TAPS - 1:
            return add64(S_WB, "s%d" % S_CIN2)
        if fstep // TAPS == 0:                           # first block of the body -> second block: always exists
            return add64(S_WB, "s%d" % S_WRAP, "s%d" % (S_WRAP + 1))
        return cond_next() + [Op("s_cselect_b32 s%d, s%d, s%d" % (S_T2, S_WRAP, S_WRAPL), "S"),
                              Op("s_cselect_b32 s%d, s%d, s%d" % (S_T3, S_WRAP + 1, S_WRAPL + 1), "S")] + \
            add64(S_WB, "s%d" % S_T2, "s%d" % S_T3)

    def h_advance(step):       # after the LAST halo piece of a refill group (third == 2): move that slot's base to its next block
        slot, third, cond = HALO_STEPS[step]             # cond: the block fetched next is the first of the NEXT body: only if one follows
        assert third == c.NT - 1
        if not cond:
            return add64(S_X[slot], "64")
        return cond_next() + [Op("s_cselect_b32 s%d, 64, 0" % S_T2, "S")] + add64(S_X[slot], "s%d" % S_T2)

    def dma_issue(piece):
        m0w, d = piece
        ops.append(m0w)
        emit("s_nop 0")
        ops.append(d)
        vm.append(d.meta["tag"])

    def vm_need(tag):
        """wait until every piece tagged `tag` has landed (pieces retire in order)"""
        if tag in vm:
            idx = len(vm) - 1 - vm[::-1].index(tag)
            n = len(vm) - 1 - idx
            assert n < 64
            emit("s_waitcnt vmcnt(%d)" % n, "W", dict(vm=n))
            del vm[: idx + 1]

    lab = lambda n: emit(".L%s_%s_%%=:" % (c.tag, n), "L", dict(name=n))
    ref = lambda n: ".L%s_%s_%%=" % (c.tag, n)

    # ------------------------------------------------------------------------------------------------ prologue
    emit("s_mov_b64 s[%d:%d], %s" % (S_WB, S_WB + 1, OPN["wbase"]), "S")
    for d in range(c.NSLOT):
        emit("s_mov_b64 s[%d:%d], %s" % (S_X[d], S_X[d] + 1, OPN["xb%d" % d]), "S")
    emit("s_mov_b32 s%d, %s" % (S_CIN2, OPN["cin2"]), "S")
    emit("s_mov_b32 s%d, %s" % (S_NBODY, OPN["nbody"]), "S")
    emit("s_mov_b32 s%d, %s" % (S_WDST, OPN["wdst"]), "S")
    emit("s_mov_b32 s%d, %s" % (S_HDST, OPN["hdst"]), "S")
    emit("s_mov_b32 s%d, %s" % (S_HDST5, OPN["hdst5"]), "S")
    emit("s_mov_b32 s%d, 0" % S_IT, "S")
    emit("s_mul_i32 s%d, s%d, 26" % (S_TMP, S_CIN2), "S")                  # 26 cin2 < 2^31
    emit("s_sub_u32 s%d, 0, s%d" % (S_WRAPL, S_TMP), "S")                  # -26 cin2, sign-extended to 64 bits
    emit("s_mov_b32 s%d, -1" % (S_WRAPL + 1), "S")
    emit("s_sub_u32 s%d, 64, s%d" % (S_WRAP, S_TMP), "S")                  # 64 - 26 cin2 < 0 for cin2 >= 64
    emit("s_mov_b32 s%d, -1" % (S_WRAP + 1), "S")
    for k in range(NS):
        emit("v_add_u32_e32 %s, %d, %s" % (vr(c.VY + k), c.W_BASE + k * c.W_STAGE, OPN["yb"]), "X")
    for r in range(c.NACC):
        emit("v_accvgpr_write_b32 %s, 0" % ar(r))
    if c.GN:
        emit("s_mov_b64 s[%d:%d], %s" % (S_G, S_G + 1, OPN["gbase"]), "S")
        emit("s_mov_b32 s%d, 0xffff0000" % S_MASK, "S")
        emit("s_mov_b32 s%d, 0xbfb8aa3b" % S_NL2E, "S")
        for o in table_load(False)[2:]:                                    # channel block 0's rows: no advance
            ops.append(o)
            vm.append(o.meta["tag"])
        for p in w_pieces(0):
            dma_issue(p)
        for o in w_advance(0):
            ops.append(o)
        # halo frames 0, 1 of channel block 0: all 12 pieces of this wave are fetched at once (ONE memory latency per tile) into
        # the fragment registers, then transformed and written in order
        pro = [(slot, k) for slot in (0, 1) for k in range(6)]
        assert 4 * len(pro) <= 64
        for n, (slot, k) in enumerate(pro):
            o = g_load(slot, k, ("pro", slot, k), pro=n)
            ops.append(o)
            vm.append(o.meta["tag"])
            if k == 5:
                ops.extend(add64(S_X[slot], "64"))
        for n, (slot, k) in enumerate(pro):
            vm_need(("G", "pro", slot, k))
            for o in xform(slot, ("pro", slot, k), ("pro", n), c.V0 + 4 * n):
                ops.append(o)
                if o.kind == "Wd":
                    pend.append(o.meta["tag"])
        emit("s_waitcnt vmcnt(0) lgkmcnt(0)", "W", dict(vm=0, lgkm=0))
        del vm[:]
        del pend[:]
        emit("s_barrier", "B")
    else:
        for slot in (0, 1):                                                # halo frames 0, 1 of channel block 0
            for third in range(c.NT):
                for p in h_pieces(slot, third, ("H", slot, "pro")):
                    dma_issue(p)
            for o in add64(S_X[slot], "64"):
                ops.append(o)
        for p in w_pieces(0):                                              # the prologue's own fragment reads need step 0's weights
            dma_issue(p)
        for o in w_advance(0):
            ops.append(o)
        emit("s_waitcnt vmcnt(0)", "W", dict(vm=0))
        del vm[:]
        emit("s_barrier", "B")
    for f in range(1, LEAD):                                               # the pieces a steady-state body top finds in flight
        for p in w_pieces(f):
            dma_issue(p)
        for o in w_advance(f):
            ops.append(o)
    for i in range(NB):
        a_read(0, i)
    for j in range(c.NBJ - 1):
        b_read(0, j)

    # ------------------------------------------------------------------------------------------------ body
    lab("body")
    emit("s_waitcnt lgkmcnt(0)", "W", dict(lgkm=0))                        # canonical state at the loop top (once per 54 steps)
    del pend[:]
    vm_top = list(vm)
    publish = {}                           # GN form: step -> halo write tags that must have retired before its barrier
    gn_fill = {}                           # GN form: step -> MFMA index -> fillers (transform, LDS write, table load)
    if c.GN:
        at = lambda time: (int(time), 2 + int((time - int(time)) * 60))
        put = lambda time, item: gn_fill.setdefault(at(time)[0], {}).setdefault(at(time)[1], []).append(item)
        nr = 0
        for start, slot, t0, t1, _ in c.GN_GROUPS:
            for i in range(6):                                   # piece i = the i-th load of the group (step start + i // 2)
                key = (start + i // 2, slot, i)
                r, nr = nr % c.NRING, nr + 1
                ring_pre[key] = r
                xo = xform(slot, key, r)
                a, b = t0 + (t1 - t0) * i / 6.0, t0 + (t1 - t0) * (i + 1) / 6.0
                assert a >= start + i // 2 + 1, "a piece is transformed less than a step behind its load"
                put(a, ("vmwait", [("T", 3), ("G",) + key]))
                for n, o in enumerate(xo):
                    if o.kind == "Wd":
                        put(a + (b - a) * (n + 0.5) / len(xo), ("dswrite", o))
                        if at(a + (b - a) * (n + 0.5) / len(xo))[0] + 1 < BODY:      # (the body's last step: lgkmcnt(0) at the loop top / exit)
                            publish.setdefault(at(a + (b - a) * (n + 0.5) / len(xo))[0] + 1, []).append(o.meta["tag"])
                    else:
                        put(a + (b - a) * (n + 0.5) / len(xo), ("valu", o))
        for time, cond in c.TAB_TIMES.items():
            tl = table_load(cond)
            put(time, ("salu", [o for o in tl if o.kind != "G"]))
            for q, o in enumerate(o for o in tl if o.kind == "G"):
                put(time + (q + 1) / 60.0, ("vmem", o))
    for s in range(BODY):
        if s % c.BAR == 0:
            vm_need(("W", s + c.BAR))      # every step whose fragment reads are issued before the next barrier
            for tag in publish.pop(s, []):
                need(tag)
            emit("s_barrier", "B", dict(step=s))
        # fillers by MFMA index
        fill = [[] for _ in range(c.NM)]
        for i in range(NB):
            fill[i].append(("a", s + 1, i))
        fill[1].append(("b", s, c.NBJ - 1))
        for j in range(c.NBJ - 1):
            fill[8 * j + 9].append(("b", s + 1, j))
        pieces = [("w", p) for p in w_pieces(s + LEAD)]
        if s in HALO_STEPS:
            slot, third = HALO_STEPS[s][:2]
            if c.GN:
                pieces += [("g", g_load(slot, k, (s, slot, k))) for k in (2 * third, 2 * third + 1)]
            else:
                pieces += [("h", p) for p in h_pieces(slot, third, ("H", slot, s))]
        gap = c.NM // max(len(pieces), 4)
        for n, (kind, p) in enumerate(pieces):
            if kind == "g":
                fill[4 + n * gap].append(("vmem", p))
            else:
                fill[3 + n * gap].append(("m0", p[0]))
                fill[4 + n * gap].append(("dma", p[1]))
            if kind == "w" and n == c.NWP - 1:
                fill[4 + n * gap].append(("salu", w_advance(s + LEAD)))
            if kind in "hg" and n == len(pieces) - 1 and HALO_STEPS[s][1] == c.NT - 1:
                fill[4 + n * gap].append(("salu", h_advance(s)))
        for key_, items in gn_fill.get(s, {}).items():
            fill[key_].extend(items)
        m = 0
        for j in range(c.NBJ):
            for i in range(NB):
                need(("A", s, i))
                need(("B", s, j))
                mfma(s, j, i)
                for f in fill[m]:
                    if f[0] == "a":
                        a_read(f[1], f[2])
                    elif f[0] == "b":
                        b_read(f[1], f[2])
                    elif f[0] == "m0":
                        ops.append(f[1])
                    elif f[0] in ("dma", "vmem"):
                        ops.append(f[1])
                        vm.append(f[1].meta["tag"])
                    elif f[0] == "vmwait":
                        for tag in f[1]:
                            vm_need(tag)
                    elif f[0] == "valu":
                        ops.append(f[1])
                    elif f[0] == "dswrite":
                        ops.append(f[1])
                        pend.append(f[1].meta["tag"])
                    else:
                        ops.extend(f[1])
                m += 1
    # loop control: the reads issued for "step 54" are the next body's step 0 (same registers, same offsets)
    emit("s_add_u32 s%d, s%d, 1" % (S_IT, S_IT), "S")
    emit("s_cmp_lt_u32 s%d, s%d" % (S_IT, S_NBODY), "S")
    emit("s_cbranch_scc1 %s" % ref("body"), "J", dict(target="body"))
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)", "W", dict(vm=0, lgkm=0))
    emit("s_nop 15")
    emit("s_nop 15")
    # both entries of the body (prologue, back edge) must find the same pieces in flight BEHIND the ones its first wait retires
    # (older pieces -- the last halo refill of the previous body -- are retired by that wait on either path): same vmcnt immediates
    def behind_first_wait(tags):
        norm = [(t[0], t[1] % BODY) if t[0] == "W" else t for t in tags]
        last = len(norm) - 1 - norm[::-1].index(("W", c.BAR))
        return norm[last + 1:]
    assert behind_first_wait(vm) == behind_first_wait(vm_top), (vm, vm_top)
    assert not publish and (not c.GN or nring[0] % c.NRING == 0)
    return ops


def body_lines(c):
    return [o.text if o.kind == "L" else "  " + o.text for o in generate(c)]


def clobbers(c):
    return ['"v%d"' % i for i in range(c.V0, c.V0 + c.VN)] + ['"a%d"' % i for i in range(c.NACC)] + \
           ['"s%d"' % i for i in range(S_FIRST, c.S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "open_sora_amd", "csrc"))
    ap.add_argument("--gn-exp", default="", help='experiments (tools/make_conv_gn_variants.sh): JSON for GN_EXP, e.g. {"valu": "notrans"}')
    args = ap.parse_args()
    if args.gn_exp:
        import json
        exp = json.loads(args.gn_exp)
        if "windows" in exp:       # {"f2" | "plain": {start: [slot, T0, T1]}}
            exp["windows"] = {k == "f2": {int(st): tuple(v) for st, v in d.items()} for k, d in exp["windows"].items()}
        GN_EXP.update(exp)
    c = Cfg(8, f2=True)
    with open(os.path.join(args.out, "convswf_body_n128.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_conv_sw_asm.py -- do not edit.  Sliding-window CausalConv3d K loop, two-frame form: 16 x 16 voxel "
                "brick of 2 frames x 128 channels, two 32-channel blocks x 27 taps per body.\n")
        for ln in body_lines(c):
            f.write('"%s\\n"\n' % ln)
    for up in (False, True):
        for nbj in (8, 4):
            c = Cfg(nbj, up)
            with open(os.path.join(args.out, "convsw%s_body_n%d.inc" % ("u" if up else "", c.BN)), "w") as f:
                f.write("// GENERATED by tools/gen_conv_sw_asm.py -- do not edit.  Sliding-window CausalConv3d K loop%s: 16 x 16 voxel brick x "
                        "%d channels, two 32-channel blocks x 27 taps per body.\n" % (" (nearest 2x upsample folded in)" if up else "", c.BN))
                for ln in body_lines(c):
                    f.write('"%s\\n"\n' % ln)
    for f2 in (False, True):
        c = Cfg(8, f2=f2, gn=True)
        with open(os.path.join(args.out, "convswg%s_body_n%d.inc" % ("f" if f2 else "", c.BN)), "w") as f:
            f.write("// GENERATED by tools/gen_conv_sw_asm.py -- do not edit.  Sliding-window CausalConv3d K loop, GN form (the input is "
                    "silu(GroupNorm(x)), applied in the halo refill)%s: 16 x 16 voxel brick x %d channels, two 32-channel blocks x 27 taps per body.\n"
                    % (", two-frame form" if f2 else "", c.BN))
            for ln in body_lines(c):
                f.write('"%s\\n"\n' % ln)
    with open(os.path.join(args.out, "convsw_regs.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_conv_sw_asm.py -- do not edit.\n")
        for up in (False, True):
            for nbj in (8, 4):
                c = Cfg(nbj, up)
                P = "OSKSW%s%d" % ("U" if up else "", c.BN)
                f.write("#define %s_SMEM %d\n#define %s_SLOT %d\n" % (P, c.SMEM, P, c.SLOT))
        c = Cfg(8, f2=True)
        f.write("#define OSKSWF128_SMEM %d\n#define OSKSWF128_SLOT %d\n" % (c.SMEM, c.SLOT))
        for nbj in (8, 4):                       # the register footprint does not depend on the halo geometry
            c = Cfg(nbj)
            f.write("#define OSKSW%d_CLOBBERS %s\n" % (c.BN, ", ".join(clobbers(c))))
        for f2 in (False, True):                 # GN form: + the register ring (6 / 8 quads), 16 scale / shift registers, 8 temporaries
            c = Cfg(8, f2=f2, gn=True)
            f.write("#define OSKSWG%s_CLOBBERS %s\n" % ("F128" if f2 else "256", ", ".join(clobbers(c))))


if __name__ == "__main__":
    main()
