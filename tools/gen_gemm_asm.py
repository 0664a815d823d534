#!/usr/bin/env python
"""Generator of the hand-scheduled gfx950 K-loop of the large-tile bf16 GEMM (open_sora_amd/csrc/gemm256.hip).

Tile 256 (M, activation rows) x BN (N, weight rows) x 64 (K); 8 waves = 512 threads (two per SIMD) in a WM x WN
grid, each wave TM x TN v_mfma_f32_32x32x16_bf16 tiles with the operands swapped as in gemm_bf16.hip (A operand =
weight fragment, B operand = activation fragment: a lane of the accumulator owns one output row m).
  BN = 256: waves 2 x 4, wave tile 128 x 64  (TM 4, TN 2, 128 accumulator AGPRs)
  BN = 128: waves 4 x 2, wave tile  64 x 64  (TM 2, TN 2,  64 accumulator AGPRs)
LDS: two stages of [256][64] activations + [BN][64] weights, 128-byte rows, source-side XOR swizzle, filled by
LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, 8 / 6 instructions per wave and K step).
One K step of a wave (after the barrier that published stage `cur`):
  fragment reads of k-sub-step 0 | the LAST k-sub-step of the previous K step (fragments already in registers)
  with the LDS-DMA of stage cur^1 in its shadows | k-sub-steps 0..2, each prefetching the fragments of the next
  into the other fragment set | vmcnt(0) lgkmcnt(0), barrier.
Accumulators live in AGPRs a[0 : 16 TM TN); the wrapper reads them out for the fused epilogue.
"""
import argparse
import os

S_AB, S_WB, S_NK, S_ADST, S_WDST, S_T, S_KL, S_TMP, S_STEP = 40, 42, 44, 45, 46, 47, 48, 49, 50
S_FIRST, S_LAST = 40, 51
OPERANDS = ["faA0", "faA1", "faA2", "faA3", "faW0", "faW1", "faW2", "faW3", "aoff0", "aoff1", "aoff2", "aoff3",
            "woff0", "woff1", "woff2", "woff3", "abase", "wbase", "nk", "adst", "wdst"]
OP = {n: "%%%d" % i for i, n in enumerate(OPERANDS)}


class Cfg:
    def __init__(self, bn, fp8=False):
        self.BN = bn
        self.FP8 = fp8
        self.TM, self.TN = (4, 2) if bn == 256 else (2, 2)
        self.NA, self.NW = 4, bn // 64           # LDS-DMA instructions per wave and stage (A: 32 / 8, W: BN/8 / 8)
        self.A_STAGE, self.W_STAGE = 32768, bn * 128
        self.W_BASE = 65536
        self.SMEM = self.W_BASE + 2 * self.W_STAGE
        self.NACC = 16 * self.TM * self.TN
        self.NFRAG = self.TM + self.TN
        # first asm-owned VGPR: two fragment sets.  fp8: a fragment is 32 bytes (8 registers), and with 128
        # accumulator AGPRs of the 256 registers a wave may own (two waves per SIMD) the VGPR half ends at v127
        self.FW = 8 if fp8 else 4
        self.V0 = (32 if bn == 256 else 64) if fp8 else 64
        self.VN = 2 * self.NFRAG * self.FW
        self.tag = ("q%d" if fp8 else "g%d") % bn

    def frag(self, fset, idx):                   # idx: 0..TM-1 activation fragments, TM.. weight fragments
        return self.V0 + (fset * self.NFRAG + idx) * self.FW


def vr(b, n=1):
    return "v%d" % b if n == 1 else "v[%d:%d]" % (b, b + n - 1)


def ar(b, n=1):
    return "a%d" % b if n == 1 else "a[%d:%d]" % (b, b + n - 1)


def gen(c):
    L = []
    e = lambda t: L.append("  " + t)
    lab = lambda n: L.append(".L%s_%s_%%=:" % (c.tag, n))   # %= : unique per emitted copy of the asm statement

    def reads(stage, ks, fset):
        out = []
        for tm in range(c.TM):
            out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, tm), 4), OP["faA%d" % ks], stage * c.A_STAGE + tm * 4096))
        for tn in range(c.TN):
            out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, c.TM + tn), 4), OP["faW%d" % ks], stage * c.W_STAGE + tn * 4096))
        return out

    def mfmas(fset):
        out = []
        for tn in range(c.TN):
            for tm in range(c.TM):
                acc = ar((tn * c.TM + tm) * 16, 16)
                out.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, vr(c.frag(fset, c.TM + tn), 4), vr(c.frag(fset, tm), 4), acc))
        return out

    def dma(stage):
        """(m0 write, dma) pairs of this wave for one stage"""
        out = []
        for i in range(c.NA):
            out.append(("s_add_u32 m0, s%d, %d" % (S_ADST, stage * c.A_STAGE + i * 8192),
                        "global_load_lds_dwordx4 %s, s[%d:%d]" % (OP["aoff%d" % i], S_AB, S_AB + 1)))
        for i in range(c.NW):
            out.append(("s_add_u32 m0, s%d, %d" % (S_WDST, stage * c.W_STAGE + i * 8192),
                        "global_load_lds_dwordx4 %s, s[%d:%d]" % (OP["woff%d" % i], S_WB, S_WB + 1)))
        return out

    def advance():
        # point both loaders at the next K step; past the last one they stay (harmless re-fetch)
        return ["s_add_u32 s%d, s%d, 1" % (S_TMP, S_KL),
                "s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NK),
                "s_cselect_b32 s%d, 128, 0" % S_STEP,
                "s_cselect_b32 s%d, s%d, s%d" % (S_KL, S_TMP, S_KL),
                "s_add_u32 s%d, s%d, s%d" % (S_AB, S_AB, S_STEP),
                "s_addc_u32 s%d, s%d, 0" % (S_AB + 1, S_AB + 1),
                "s_add_u32 s%d, s%d, s%d" % (S_WB, S_WB, S_STEP),
                "s_addc_u32 s%d, s%d, 0" % (S_WB + 1, S_WB + 1)]

    def interleave(mf, fill):
        """after MFMA i emit fill[i] (a list of instructions)"""
        for i, m in enumerate(mf):
            e(m)
            for f in (fill[i] if i < len(fill) else []):
                e(f)

    # ---- setup
    e("s_mov_b64 s[%d:%d], %s" % (S_AB, S_AB + 1, OP["abase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_WB, S_WB + 1, OP["wbase"]))
    e("s_mov_b32 s%d, %s" % (S_NK, OP["nk"]))
    e("s_mov_b32 s%d, %s" % (S_ADST, OP["adst"]))
    e("s_mov_b32 s%d, %s" % (S_WDST, OP["wdst"]))
    e("s_mov_b32 s%d, 0" % S_T)
    e("s_mov_b32 s%d, 0" % S_KL)
    for r in range(c.NACC):
        e("v_accvgpr_write_b32 %s, 0" % ar(r))
    # ---- prologue: stage 0, then what step 0 does before its entry point (DMA of stage 1, reads of k-sub-step 0)
    for m0w, d in dma(0):
        e(m0w); e("s_nop 0"); e(d)
    for a in advance():
        e(a)
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    for m0w, d in dma(1):
        e(m0w); e("s_nop 0"); e(d)
    for a in advance():
        e(a)
    for r in reads(0, 0, 0):
        e(r)
    e("s_branch .L%s_entry0_%%=" % c.tag)

    for k in range(2):
        cur = k
        lab("step%d" % k)
        for r in reads(cur, 0, 0):
            e(r)
        # trailing k-sub-step 3 of the previous K step (fragment set 1) with the LDS-DMA of stage cur^1 in its shadows
        mf = mfmas(1)
        pieces = dma(cur ^ 1)
        fill = [[] for _ in mf]
        # M0 write before MFMA i, DMA after it: realised by putting the M0 write at the end of shadow i-1
        e(pieces[0][0])
        for i in range(len(mf)):
            if i < len(pieces):
                fill[i].append(pieces[i][1])
                if i + 1 < len(pieces):
                    fill[i].append(pieces[i + 1][0])
        rest = pieces[len(mf):]
        interleave(mf, fill)
        for m0w, d in rest:
            e(m0w); e("s_nop 0"); e(d)
        for a in advance():
            e(a)
        lab("entry%d" % k)
        for ks in range(3):
            fset = ks % 2
            e("s_waitcnt lgkmcnt(0)")
            mf = mfmas(fset)
            rd = reads(cur, ks + 1, fset ^ 1)
            fill = [[] for _ in mf]
            for i, r in enumerate(rd):
                fill[i].append(r)
            interleave(mf, fill)
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e("s_barrier")
        e("s_add_u32 s%d, s%d, 1" % (S_T, S_T))
        e("s_cmp_lt_u32 s%d, s%d" % (S_T, S_NK))
        e("s_cbranch_scc0 .L%s_exit_%%=" % c.tag)
        if k == 1:
            e("s_branch .L%s_step0_%%=" % c.tag)
    lab("exit")
    for m in mfmas(1):
        e(m)
    e("s_nop 15")
    e("s_nop 15")
    return L


PERS_OPERANDS = ["faA0", "faW0", "aoff0", "aoff1", "aoff2", "aoff3", "woff0", "woff1", "woff2", "woff3",
                 "aoffn0", "aoffn1", "aoffn2", "aoffn3", "woffn0", "woffn1", "woffn2", "woffn3", "boff",
                 "abase", "wbase", "bias", "nk", "adst", "wdst", "flags"]
POP = {n: "%%%d" % i for i, n in enumerate(PERS_OPERANDS)}
S_PFLAGS, S_PBIAS = 52, 54          # flags; bias base (pair)
PERS_S_LAST = 55
PF_PREFETCHED, PF_HAS_NEXT, PF_BIAS = 0, 1, 2   # flag bits


def gen_pers(c, sched=0):
    """K loop of gen() for a PERSISTENT workgroup (csrc/gemm256p.hip): the asm statement is executed once per output
    tile of the workgroup's tile list, and the fixed costs of a tile move off the matrix pipe's critical path:
      * exit: after the last barrier both LDS stages are free -> the LDS-DMA of the NEXT tile's K steps 0 and 1 (per-lane
        source offsets aoffn / woffn) is issued in the shadows of the trailing k-sub-step, so it lands while the wrapper
        runs this tile's epilogue (flags bit 1 = there is a next tile);
      * entry: flags bit 0 = stages 0 / 1 are already in flight (issued by the previous tile's exit): no load prologue;
      * the accumulators start from the bias (flags bit 2; 16 f32 per 32-column tile and lane, loaded once per tile
        into the idle fragment registers) instead of zero, so the epilogue of an un-gated Linear issues no load at all.
    The fragment addresses of k-sub-steps 1..3 are faX0 ^ (ks << 5) (the XOR swizzle only touches address bits 4..6)."""
    assert not c.FP8
    L = []
    e = lambda t: L.append("  " + t)
    lab = lambda n: L.append(".Lp%s_%s_%%=:" % (c.tag, n))
    ref = lambda n: ".Lp%s_%s_%%=" % (c.tag, n)
    VX = c.V0 + c.VN                       # 6 derived fragment addresses: A ks 1..3, W ks 1..3
    fa = {("A", 0): POP["faA0"], ("W", 0): POP["faW0"]}
    for ks in range(1, 4):
        fa[("A", ks)] = vr(VX + ks - 1)
        fa[("W", ks)] = vr(VX + 3 + ks - 1)

    def reads(stage, ks, fset):
        out = []
        for tm in range(c.TM):
            out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, tm), 4), fa[("A", ks)], stage * c.A_STAGE + tm * 4096))
        for tn in range(c.TN):
            out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, c.TM + tn), 4), fa[("W", ks)], stage * c.W_STAGE + tn * 4096))
        return out

    def mfmas(fset):
        out = []
        for tn in range(c.TN):
            for tm in range(c.TM):
                acc = ar((tn * c.TM + tm) * 16, 16)
                out.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, vr(c.frag(fset, c.TM + tn), 4), vr(c.frag(fset, tm), 4), acc))
        return out

    def dma(stage, nxt=False):
        sfx = "n" if nxt else ""
        out = []
        for i in range(c.NA):
            out.append(("s_add_u32 m0, s%d, %d" % (S_ADST, stage * c.A_STAGE + i * 8192),
                        "global_load_lds_dwordx4 %s, s[%d:%d]" % (POP["aoff%s%d" % (sfx, i)], S_AB, S_AB + 1)))
        for i in range(c.NW):
            out.append(("s_add_u32 m0, s%d, %d" % (S_WDST, stage * c.W_STAGE + i * 8192),
                        "global_load_lds_dwordx4 %s, s[%d:%d]" % (POP["woff%s%d" % (sfx, i)], S_WB, S_WB + 1)))
        return out

    def advance():
        return ["s_add_u32 s%d, s%d, 1" % (S_TMP, S_KL),
                "s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NK),
                "s_cselect_b32 s%d, 128, 0" % S_STEP,
                "s_cselect_b32 s%d, s%d, s%d" % (S_KL, S_TMP, S_KL),
                "s_add_u32 s%d, s%d, s%d" % (S_AB, S_AB, S_STEP),
                "s_addc_u32 s%d, s%d, 0" % (S_AB + 1, S_AB + 1),
                "s_add_u32 s%d, s%d, s%d" % (S_WB, S_WB, S_STEP),
                "s_addc_u32 s%d, s%d, 0" % (S_WB + 1, S_WB + 1)]

    def plain(pieces):
        for m0w, d in pieces:
            e(m0w); e("s_nop 0"); e(d)

    # ---- setup
    e("s_mov_b64 s[%d:%d], %s" % (S_AB, S_AB + 1, POP["abase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_WB, S_WB + 1, POP["wbase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_PBIAS, S_PBIAS + 1, POP["bias"]))
    e("s_mov_b32 s%d, %s" % (S_NK, POP["nk"]))
    e("s_mov_b32 s%d, %s" % (S_ADST, POP["adst"]))
    e("s_mov_b32 s%d, %s" % (S_WDST, POP["wdst"]))
    e("s_mov_b32 s%d, %s" % (S_PFLAGS, POP["flags"]))
    e("s_mov_b32 s%d, 0" % S_T)
    e("s_mov_b32 s%d, 0" % S_KL)
    for ks in range(1, 4):
        e("v_xor_b32_e32 %s, %d, %s" % (fa[("A", ks)], ks << 5, POP["faA0"]))
        e("v_xor_b32_e32 %s, %d, %s" % (fa[("W", ks)], ks << 5, POP["faW0"]))
    # ---- entry: K step 0 -> stage 0 unless the previous tile's exit already fetched it
    e("s_bitcmp1_b32 s%d, %d" % (S_PFLAGS, PF_PREFETCHED))
    e("s_cbranch_scc1 %s" % ref("have0"))
    plain(dma(0))
    lab("have0")
    for a in advance():
        e(a)
    # bias of this lane's accumulator columns -> the idle fragment registers (consumed before the first fragment read)
    e("s_bitcmp1_b32 s%d, %d" % (S_PFLAGS, PF_BIAS))
    e("s_cbranch_scc0 %s" % ref("nobias"))
    for tn in range(c.TN):
        for qd in range(4):
            e("global_load_dwordx4 %s, %s, s[%d:%d] offset:%d" % (vr(c.V0 + (tn * 4 + qd) * 4, 4), POP["boff"], S_PBIAS, S_PBIAS + 1,
                                                                   (tn * 32 + qd * 8) * 4))
    lab("nobias")
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    e("s_bitcmp1_b32 s%d, %d" % (S_PFLAGS, PF_PREFETCHED))
    e("s_cbranch_scc1 %s" % ref("have1"))
    plain(dma(1))
    lab("have1")
    if sched == 0:
        for a in advance():
            e(a)
    else:
        # schedules 1 / 2 enter the loop at entry0, whose first shadows carry the LAST LDS-DMA pieces of the stage-1 fetch
        # (re-issued here: same bytes to the same LDS rows, harmless) followed by the pointer advance: set up M0 for them
        pcs = dma(1)
        n_trail = (c.TM * c.TN + 1) // 2 if sched == 2 else c.TM * c.TN - (c.NFRAG + 1) // 2
        e(pcs[n_trail][0])
    e("s_bitcmp1_b32 s%d, %d" % (S_PFLAGS, PF_BIAS))
    e("s_cbranch_scc0 %s" % ref("zero"))
    for tn in range(c.TN):
        for tm in range(c.TM):
            for r in range(16):
                e("v_accvgpr_write_b32 %s, %s" % (ar((tn * c.TM + tm) * 16 + r), vr(c.V0 + tn * 16 + r)))
    e("s_branch %s" % ref("inited"))
    lab("zero")
    for r in range(c.NACC):
        e("v_accvgpr_write_b32 %s, 0" % ar(r))
    lab("inited")
    e("s_nop 1")
    for r in reads(0, 0, 0):
        e(r)
    e("s_branch %s" % ref("entry0"))

    def step_sched0(cur):
        for r in reads(cur, 0, 0):
            e(r)
        mf = mfmas(1)
        pieces = dma(cur ^ 1)
        e(pieces[0][0])
        for i, m in enumerate(mf):
            e(m)
            if i < len(pieces):
                e(pieces[i][1])
                if i + 1 < len(pieces):
                    e(pieces[i + 1][0])
        plain(pieces[len(mf):])
        for a in advance():
            e(a)
        lab("entry%d" % cur)
        for ks in range(3):
            fset = ks % 2
            e("s_waitcnt lgkmcnt(0)")
            mf = mfmas(fset)
            rd = reads(cur, ks + 1, fset ^ 1)
            for i, m in enumerate(mf):
                e(m)
                if i < len(rd):
                    e(rd[i])

    def step_sched1(cur):
        """matrix pipe first: the trailing k-sub-step (fragments in registers) starts right behind the barrier, the
        fragment reads of the new stage ride in its first shadows TWO per shadow (they have the rest of the trailing
        sub-step to land), then one LDS-DMA piece per shadow; every later sub-step also issues its prefetch reads two per
        shadow at its head, so a fragment has most of a sub-step of slack before the lgkmcnt wait that guards it."""
        mf = mfmas(1)
        rd = reads(cur, 0, 0)
        pieces = dma(cur ^ 1)
        npair = (len(rd) + 1) // 2
        slots = [[] for _ in range(len(mf) + c.TM * c.TN)]      # shadows of the trailing sub-step, then of sub-step 0
        for i in range(npair):
            slots[i] += rd[2 * i: 2 * i + 2]
        e(pieces[0][0])
        for j, (m0w, d) in enumerate(pieces):
            slots[npair + j].append(d)
            if j + 1 < len(pieces):
                slots[npair + j].append(pieces[j + 1][0])
        for i, m in enumerate(mf):
            e(m)
            for x in slots[i]:
                e(x)
        lab("entry%d" % cur)
        carry = slots[len(mf):]                                 # LDS-DMA pieces that did not fit behind the reads
        for ks in range(3):
            fset = ks % 2
            e("s_waitcnt lgkmcnt(0)")
            mf = mfmas(fset)
            rd = reads(cur, ks + 1, fset ^ 1)
            sh = [[] for _ in mf]
            first = 0
            if ks == 0:
                for i, x in enumerate(carry):
                    if x:
                        sh[i] += x
                        first = i + 1
            for i in range((len(rd) + 1) // 2):
                sh[min(first + i, len(mf) - 1)] += rd[2 * i: 2 * i + 2]
            for i, m in enumerate(mf):
                e(m)
                for x in sh[i]:
                    e(x)
            if ks == 0:
                for a in advance():     # only now: the carried LDS-DMA pieces above still address THIS fetch's K step
                    e(a)

    def step_sched2(cur):
        """schedule 1 with the LDS-DMA pieces one per TWO shadows over the trailing sub-step and sub-step 0 (see gen_w4:
        an LDS-DMA instruction holds the issue port for several MFMA shadows)"""
        NM = c.TM * c.TN
        pieces = dma(cur ^ 1)
        dma_slots = [2 * j for j in range(len(pieces))]
        assert dma_slots[-1] < 2 * NM
        slots = [[] for _ in range(4 * NM)]
        e(pieces[0][0])
        for j, (m0w, d) in enumerate(pieces):
            slots[dma_slots[j]].append(d)
            if j + 1 < len(pieces):
                slots[dma_slots[j]].append(pieces[j + 1][0])
        last = dma_slots[-1]
        for blk, (ks, fset) in enumerate(((0, 0), (1, 1), (2, 0), (3, 1))):
            rd = reads(cur, ks, fset)
            free = [i for i in range(blk * NM, (blk + 1) * NM - 1) if i not in dma_slots]
            for i in range((len(rd) + 1) // 2):
                slots[free[i]] += rd[2 * i: 2 * i + 2]
        for blk in range(4):
            if blk == 1:
                lab("entry%d" % cur)
            if blk >= 1:
                e("s_waitcnt lgkmcnt(0)")
            mf = mfmas(1 if blk % 2 == 0 else 0)
            for i, m in enumerate(mf):
                e(m)
                for x in slots[blk * NM + i]:
                    e(x)
                if blk * NM + i == last:
                    for a in advance():
                        e(a)

    for k in range(2):
        cur = k
        lab("step%d" % k)
        (step_sched2 if sched == 2 else (step_sched1 if sched == 1 else step_sched0))(cur)
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e("s_barrier")
        e("s_add_u32 s%d, s%d, 1" % (S_T, S_T))
        e("s_cmp_lt_u32 s%d, s%d" % (S_T, S_NK))
        e("s_cbranch_scc0 %s" % ref("exit"))
        if k == 1:
            e("s_branch %s" % ref("step0"))
    lab("exit")
    # both stages are free now (every fragment of the last K step is in registers): fetch the next tile's first two K steps
    e("s_bitcmp1_b32 s%d, %d" % (S_PFLAGS, PF_HAS_NEXT))
    e("s_cbranch_scc0 %s" % ref("last"))
    e("s_mov_b64 s[%d:%d], %s" % (S_AB, S_AB + 1, POP["abase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_WB, S_WB + 1, POP["wbase"]))
    mf = mfmas(1)
    pieces = dma(0, nxt=True)
    e(pieces[0][0])
    for i, m in enumerate(mf):
        e(m)
        if i < len(pieces):
            e(pieces[i][1])
            if i + 1 < len(pieces):
                e(pieces[i + 1][0])
    plain(pieces[len(mf):])
    e("s_cmp_gt_u32 s%d, 1" % S_NK)                      # K step 1 exists?  (else stage 1 re-fetches step 0: harmless)
    e("s_cselect_b32 s%d, 128, 0" % S_STEP)
    e("s_add_u32 s%d, s%d, s%d" % (S_AB, S_AB, S_STEP))
    e("s_addc_u32 s%d, s%d, 0" % (S_AB + 1, S_AB + 1))
    e("s_add_u32 s%d, s%d, s%d" % (S_WB, S_WB, S_STEP))
    e("s_addc_u32 s%d, s%d, 0" % (S_WB + 1, S_WB + 1))
    plain(dma(1, nxt=True))
    e("s_branch %s" % ref("end"))
    lab("last")
    for m in mfmas(1):
        e(m)
    lab("end")
    e("s_nop 15")
    e("s_nop 15")
    return L


class Cfg4:
    """4 waves (one per SIMD, the whole 512-register file each), 2 x 2 over a 256 x 256 x 64 tile: wave tile 128 x 128 =
    4 x 4 MFMA tiles in 256 accumulator AGPRs.  A k-sub-step reads 8 fragments for 16 MFMAs (the 8-wave layouts: 6 for 8):
    a third less LDS read traffic per flop, half the waves per barrier."""

    def __init__(self):
        self.BN, self.FP8 = 256, False
        self.TM, self.TN = 4, 4
        self.NA, self.NW = 8, 8                 # LDS-DMA instructions per wave and stage (32 KiB / 4 waves / 1 KiB)
        self.DMA_STRIDE = 4096                  # LDS bytes between a wave's consecutive LDS-DMA instructions (4 waves x 1 KiB)
        self.A_STAGE, self.W_STAGE = 32768, 32768
        self.W_BASE = 65536
        self.SMEM = self.W_BASE + 2 * self.W_STAGE
        self.NACC = 256
        self.NFRAG = 8
        self.FW = 4
        self.V0 = 128                           # asm-owned VGPRs: v128.. (two fragment sets = 64, then 6 addresses)
        self.VN = 2 * self.NFRAG * self.FW
        self.tag = "w256"

    def frag(self, fset, idx):
        return self.V0 + (fset * self.NFRAG + idx) * self.FW


W4_OPERANDS = ["faA0", "faW0"] + ["aoff%d" % i for i in range(8)] + ["woff%d" % i for i in range(8)] + \
              ["boff", "abase", "wbase", "bias", "nk", "adst", "wdst", "flags", "dA", "dW", "aoffp", "woffp"]
WOP = {n: "%%%d" % i for i, n in enumerate(W4_OPERANDS)}
S_DA, S_DW, S_PFA, S_PFW = 56, 57, 58, 60
S_RA, S_RW, S_KOFF = 64, 68, 72          # buffer resources of A and W (4 SGPRs each), K-step byte offset (dmak = 1)
W4_S_LAST = 72


def gen_w4(c, pf=0, abl=0, dmak=0, spread=1, rd_per=2):
    """Persistent-workgroup K loop (see gen_pers) for the 4-wave layout of Cfg4.  Differences: 16 MFMAs, 8 fragment
    reads and 16 LDS-DMA pieces per k-sub-step / K step and wave, placed "matrix pipe first" (schedule 1 of gen_pers:
    behind a barrier the trailing sub-step starts at once, reads ride two per shadow at the head of a sub-step, one
    LDS-DMA piece per shadow behind them); the next tile's source offsets are this tile's plus a wave-uniform byte
    delta (dA, dW: valid when both tiles are interior -- the wrapper clears flags bit 1 otherwise), computed into the
    idle fragment registers at the exit."""
    L = []
    e = lambda t: L.append("  " + t)
    lab = lambda n: L.append(".L%s_%s_%%=:" % (c.tag, n))
    ref = lambda n: ".L%s_%s_%%=" % (c.tag, n)
    VX = c.V0 + c.VN
    fa = {("A", 0): WOP["faA0"], ("W", 0): WOP["faW0"]}
    for ks in range(1, 4):
        fa[("A", ks)] = vr(VX + ks - 1)
        fa[("W", ks)] = vr(VX + 3 + ks - 1)
    NM = c.TM * c.TN

    def reads(stage, ks, fset):
        out = []
        for tm in range(c.TM):
            out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, tm), 4), fa[("A", ks)], stage * c.A_STAGE + tm * 4096))
        for tn in range(c.TN):
            out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, c.TM + tn), 4), fa[("W", ks)], stage * c.W_STAGE + tn * 4096))
        return out

    def mfmas(fset):
        out = []
        for tn in range(c.TN):
            for tm in range(c.TM):
                acc = ar((tn * c.TM + tm) * 16, 16)
                if abl & 32:   # bit 5 (timing only): the same flops as TWO v_mfma_f32_16x16x32_bf16 on 4-register accumulators
                    b0 = (tn * c.TM + tm) * 16
                    f2 = fset ^ 1 if abl & 64 else fset      # bit 6: the second MFMA takes its operands from the other fragment set
                    out.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s\\n  v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (
                        ar(b0, 4), vr(c.frag(fset, c.TM + tn), 4), vr(c.frag(fset, tm), 4), ar(b0, 4),
                        ar(b0 + 4, 4), vr(c.frag(f2, c.TM + tn), 4), vr(c.frag(f2, tm), 4), ar(b0 + 4, 4)))
                    continue
                out.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, vr(c.frag(fset, c.TM + tn), 4), vr(c.frag(fset, tm), 4), acc))
        return out

    # dmak: which LDS-DMA instruction fetches the tiles.  0 = global_load_lds_dwordx4 (FLAT encoding: 32-bit lane offset +
    # 64-bit scalar base that the loop advances), 1 = buffer_load_dwordx4 ... offen lds (MUBUF: lane offset + buffer
    # resource + ONE scalar K offset that the loop advances) -- the form hipBLASLt's kernels use.
    def dma(stage, aregs=None, wregs=None, koff=None):
        aregs = aregs or [WOP["aoff%d" % i] for i in range(c.NA)]
        wregs = wregs or [WOP["woff%d" % i] for i in range(c.NW)]
        ko = "s%d" % S_KOFF if koff is None else koff
        out = []
        for i in range(c.NA):
            ld = ("buffer_load_dwordx4 %s, s[%d:%d], %s offen lds" % (aregs[i], S_RA, S_RA + 3, ko)) if dmak else \
                 ("global_load_lds_dwordx4 %s, s[%d:%d]" % (aregs[i], S_AB, S_AB + 1))
            out.append(("s_add_u32 m0, s%d, %d" % (S_ADST, stage * c.A_STAGE + i * c.DMA_STRIDE), ld))
        for i in range(c.NW):
            ld = ("buffer_load_dwordx4 %s, s[%d:%d], %s offen lds" % (wregs[i], S_RW, S_RW + 3, ko)) if dmak else \
                 ("global_load_lds_dwordx4 %s, s[%d:%d]" % (wregs[i], S_WB, S_WB + 1))
            out.append(("s_add_u32 m0, s%d, %d" % (S_WDST, stage * c.W_STAGE + i * c.DMA_STRIDE), ld))
        return out

    def advance():
        head = ["s_add_u32 s%d, s%d, 1" % (S_TMP, S_KL),
                "s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NK),
                "s_cselect_b32 s%d, 128, 0" % S_STEP,
                "s_cselect_b32 s%d, s%d, s%d" % (S_KL, S_TMP, S_KL)]
        if dmak:
            return head + ["s_add_u32 s%d, s%d, s%d" % (S_KOFF, S_KOFF, S_STEP)]
        return head + ["s_add_u32 s%d, s%d, s%d" % (S_AB, S_AB, S_STEP),
                       "s_addc_u32 s%d, s%d, 0" % (S_AB + 1, S_AB + 1),
                       "s_add_u32 s%d, s%d, s%d" % (S_WB, S_WB, S_STEP),
                       "s_addc_u32 s%d, s%d, 0" % (S_WB + 1, S_WB + 1)]

    def plain(pieces):
        for m0w, d in pieces:
            e(m0w); e("s_nop 0"); e(d)

    N_TRAIL = NM - (c.NFRAG + 1) // 2          # LDS-DMA pieces issued in the trailing sub-step; the rest ride in sub-step 0

    # ---- setup
    e("s_mov_b64 s[%d:%d], %s" % (S_AB, S_AB + 1, WOP["abase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_WB, S_WB + 1, WOP["wbase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_PBIAS, S_PBIAS + 1, WOP["bias"]))
    e("s_mov_b32 s%d, %s" % (S_NK, WOP["nk"]))
    e("s_mov_b32 s%d, %s" % (S_ADST, WOP["adst"]))
    e("s_mov_b32 s%d, %s" % (S_WDST, WOP["wdst"]))
    e("s_mov_b32 s%d, %s" % (S_PFLAGS, WOP["flags"]))
    e("s_mov_b32 s%d, %s" % (S_DA, WOP["dA"]))
    e("s_mov_b32 s%d, %s" % (S_DW, WOP["dW"]))
    if dmak:   # raw buffer resources: base, stride 0, num_records 2^32 - 1 bytes, gfx9 data-format word
        for rs, base in ((S_RA, WOP["abase"]), (S_RW, WOP["wbase"])):
            e("s_mov_b64 s[%d:%d], %s" % (rs, rs + 1, base))
            e("s_and_b32 s%d, s%d, 0xffff" % (rs + 1, rs + 1))
            e("s_mov_b32 s%d, -1" % (rs + 2))
            e("s_mov_b32 s%d, 0x00020000" % (rs + 3))
        e("s_mov_b32 s%d, 0" % S_KOFF)
    e("s_mov_b32 s%d, 0" % S_T)
    e("s_mov_b32 s%d, 0" % S_KL)
    for ks in range(1, 4):
        e("v_xor_b32_e32 %s, %d, %s" % (fa[("A", ks)], ks << 5, WOP["faA0"]))
        e("v_xor_b32_e32 %s, %d, %s" % (fa[("W", ks)], ks << 5, WOP["faW0"]))
    e("s_bitcmp1_b32 s%d, %d" % (S_PFLAGS, PF_PREFETCHED))
    e("s_cbranch_scc1 %s" % ref("have0"))
    plain(dma(0))
    lab("have0")
    for a in advance():
        e(a)
    e("s_bitcmp1_b32 s%d, %d" % (S_PFLAGS, PF_BIAS))
    e("s_cbranch_scc0 %s" % ref("nobias"))
    for tn in range(c.TN):
        for qd in range(4):
            e("global_load_dwordx4 %s, %s, s[%d:%d] offset:%d" % (vr(c.V0 + (tn * 4 + qd) * 4, 4), WOP["boff"], S_PBIAS, S_PBIAS + 1,
                                                                   (tn * 32 + qd * 8) * 4))
    lab("nobias")
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    e("s_bitcmp1_b32 s%d, %d" % (S_PFLAGS, PF_PREFETCHED))
    e("s_cbranch_scc1 %s" % ref("have1"))
    plain(dma(1))
    lab("have1")
    if spread > 1:
        e(dma(1)[(NM + spread - 1) // spread][0])   # first piece whose shadow lies behind entry0
    else:
        e(dma(1)[N_TRAIL][0])                  # M0 for the pieces entry0 (re-)issues before it advances the pointers
    e("s_bitcmp1_b32 s%d, %d" % (S_PFLAGS, PF_BIAS))
    e("s_cbranch_scc0 %s" % ref("zero"))
    for tn in range(c.TN):
        for tm in range(c.TM):
            for r in range(16):
                e("v_accvgpr_write_b32 %s, %s" % (ar((tn * c.TM + tm) * 16 + r), vr(c.V0 + tn * 16 + r)))
    e("s_branch %s" % ref("inited"))
    lab("zero")
    for r in range(c.NACC):
        e("v_accvgpr_write_b32 %s, 0" % ar(r))
    lab("inited")
    e("s_nop 1")
    for r in reads(0, 0, 0):
        e(r)
    e("s_branch %s" % ref("entry0"))

    def step_spread(cur):
        """spread > 1: one LDS-DMA piece every `spread` MFMA shadows from the trailing sub-step onwards (instead of one
        per shadow in 16 consecutive shadows); fragment reads two per shadow in the first free shadows of each
        sub-step.  The issue window is the trailing sub-step + sub-steps 0 and 1: the last pieces still have sub-step 2
        to land before the barrier's vmcnt(0)."""
        pieces = dma(cur ^ 1)
        nsh = 3 * NM                                            # shadows of: trailing sub-step, sub-step 0, sub-step 1
        slots = [[] for _ in range(4 * NM)]
        dma_slots = [spread * j for j in range(len(pieces))]
        assert dma_slots[-1] < nsh, "every piece must be issued before sub-step 2"
        e(pieces[0][0])
        for j, (m0w, d) in enumerate(pieces):
            slots[dma_slots[j]].append(d)
            if j + 1 < len(pieces):
                slots[dma_slots[j]].append(pieces[j + 1][0])
        last = dma_slots[-1]
        # fragment reads: sub-step g's fragments are read during the sub-step before it (index: 0 = trailing)
        for blk, (ks, fset) in enumerate(((0, 0), (1, 1), (2, 0), (3, 1))):
            rd = reads(cur, ks, fset)
            if rd_per == 1:      # one read per shadow, in the first shadows of the sub-step (a shadow may also carry an LDS-DMA piece)
                for i, r in enumerate(rd):
                    slots[blk * NM + i].insert(0, r)
                continue
            free = [i for i in range(blk * NM, (blk + 1) * NM - 3) if i not in dma_slots]
            for i in range((len(rd) + rd_per - 1) // rd_per):
                slots[free[i]] += rd[rd_per * i: rd_per * i + rd_per]
        for blk in range(4):
            if blk == 1:
                lab("entry%d" % cur)
            if blk >= 1 and not abl & 4:                        # abl bit 2: no wait for the fragment reads (timing only)
                e("s_waitcnt lgkmcnt(0)")
            mf = mfmas(1 if blk % 2 == 0 else 0)                # trailing: set 1, sub-step 0: set 0, 1: set 1, 2: set 0
            for i, m in enumerate(mf):
                e(m)
                for x in slots[blk * NM + i]:
                    if abl & 8 and x.startswith("ds_read"):      # abl bit 3: no fragment reads at all (timing only)
                        continue
                    if abl & 16 and ("global_load_lds" in x or x.startswith("s_add_u32 m0")):   # bit 4: no LDS-DMA in the loop
                        continue
                    e(x)
                if blk * NM + i == last:
                    for a in advance():
                        e(a)

    def step(cur):
        if spread > 1:
            return step_spread(cur)
        mf = mfmas(1)
        rd = reads(cur, 0, 0)
        pieces = dma(cur ^ 1)
        npair = (len(rd) + 1) // 2
        slots = [[] for _ in range(2 * NM)]
        for i in range(npair):
            slots[i] += rd[2 * i: 2 * i + 2]
        e(pieces[0][0])
        for j, (m0w, d) in enumerate(pieces):
            slots[npair + j].append(d)
            if j + 1 < len(pieces):
                slots[npair + j].append(pieces[j + 1][0])
        for i, m in enumerate(mf):
            e(m)
            for x in slots[i]:
                e(x)
        lab("entry%d" % cur)
        carry = slots[NM:]
        for ks in range(3):
            fset = ks % 2
            e("s_waitcnt lgkmcnt(0)")
            mf = mfmas(fset)
            rd = reads(cur, ks + 1, fset ^ 1)
            sh = [[] for _ in mf]
            first = 0
            if ks == 0:
                for i, x in enumerate(carry):
                    if x:
                        sh[i] += x
                        first = i + 1
            for i in range((len(rd) + 1) // 2):
                sh[min(first + i, len(mf) - 1)] += rd[2 * i: 2 * i + 2]
            for i, m in enumerate(mf):
                e(m)
                for x in sh[i]:
                    e(x)
            if ks == 0:
                for a in advance():
                    e(a)
                if pf:
                    assert not dmak
                    # L2 software prefetch: touch one dword per 128-byte line of the A / W slices of the K step AFTER the
                    # one the loaders now point at (lane = row: aoffp / woffp), two K steps before its LDS-DMA is issued,
                    # so that fetch finds its lines in L2 instead of waiting for the fabric inside the barrier's vmcnt.
                    # These two loads are the youngest VMEM operations of the step: the barrier waits vmcnt(2).
                    VD = c.V0 + c.VN + 6
                    e("s_add_u32 s%d, s%d, 1" % (S_TMP, S_KL))
                    e("s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NK))
                    e("s_cselect_b32 s%d, 128, 0" % S_STEP)
                    e("s_add_u32 s%d, s%d, s%d" % (S_PFA, S_AB, S_STEP))
                    e("s_addc_u32 s%d, s%d, 0" % (S_PFA + 1, S_AB + 1))
                    e("s_add_u32 s%d, s%d, s%d" % (S_PFW, S_WB, S_STEP))
                    e("s_addc_u32 s%d, s%d, 0" % (S_PFW + 1, S_WB + 1))
                    e("global_load_dword %s, %s, s[%d:%d]" % (vr(VD), WOP["aoffp"], S_PFA, S_PFA + 1))
                    e("global_load_dword %s, %s, s[%d:%d]" % (vr(VD + 1), WOP["woffp"], S_PFW, S_PFW + 1))

    for k in range(2):
        lab("step%d" % k)
        step(k)
        # abl (timing ablations, WRONG RESULTS, tools only): bit 0 = no LDS-DMA wait at the barrier, bit 1 = no barrier
        if abl & 1:
            e("s_waitcnt lgkmcnt(0)")
        else:
            e("s_waitcnt vmcnt(%d) lgkmcnt(0)" % (2 if pf else 0))
        if not abl & 2:
            e("s_barrier")
        e("s_add_u32 s%d, s%d, 1" % (S_T, S_T))
        e("s_cmp_lt_u32 s%d, s%d" % (S_T, S_NK))
        e("s_cbranch_scc0 %s" % ref("exit"))
        if k == 1:
            e("s_branch %s" % ref("step0"))
    lab("exit")
    if abl:
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
    e("s_bitcmp1_b32 s%d, %d" % (S_PFLAGS, PF_HAS_NEXT))
    e("s_cbranch_scc0 %s" % ref("last"))
    e("s_mov_b64 s[%d:%d], %s" % (S_AB, S_AB + 1, WOP["abase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_WB, S_WB + 1, WOP["wbase"]))
    # next tile's per-lane source offsets into fragment set 0 (its last MFMAs were issued before the barrier)
    an = [vr(c.V0 + i) for i in range(c.NA)]
    wn = [vr(c.V0 + c.NA + i) for i in range(c.NW)]
    for i in range(c.NA):
        e("v_add_u32_e32 %s, s%d, %s" % (an[i], S_DA, WOP["aoff%d" % i]))
    for i in range(c.NW):
        e("v_add_u32_e32 %s, s%d, %s" % (wn[i], S_DW, WOP["woff%d" % i]))
    mf = mfmas(1)
    pieces = dma(0, an, wn, koff="0")
    e(pieces[0][0])
    for i, m in enumerate(mf):
        e(m)
        if i < len(pieces):
            e(pieces[i][1])
            if i + 1 < len(pieces):
                e(pieces[i + 1][0])
    plain(pieces[len(mf):])
    e("s_cmp_gt_u32 s%d, 1" % S_NK)
    e("s_cselect_b32 s%d, 128, 0" % S_STEP)
    if not dmak:
        e("s_add_u32 s%d, s%d, s%d" % (S_AB, S_AB, S_STEP))
        e("s_addc_u32 s%d, s%d, 0" % (S_AB + 1, S_AB + 1))
        e("s_add_u32 s%d, s%d, s%d" % (S_WB, S_WB, S_STEP))
        e("s_addc_u32 s%d, s%d, 0" % (S_WB + 1, S_WB + 1))
    plain(dma(1, an, wn, koff="s%d" % S_STEP))
    e("s_branch %s" % ref("end"))
    lab("last")
    for m in mfmas(1):
        e(m)
    lab("end")
    e("s_nop 15")
    e("s_nop 15")
    return L

class Cfg4x:
    """Cfg4's 4 waves x (128 x 128) on v_mfma_f32_16x16x32_bf16: 8 x 8 accumulator tiles of 16 x 16 (4 registers each, the
    same 256 AGPRs), a K step = TWO sub-steps of K = 32.  A fragment is 16 rows x 32 k -- one ds_read_b128 per lane (row
    l % 16, k group l / 16) out of the SAME LDS image (rows of 128 bytes, 16-byte chunk c of row r at (c ^ ((r >> 1) & 7))):
    conflict-free for this lane map too.  Why (profiles/r02_gemm_experiments.md): with the K loop otherwise unchanged, issuing
    the same flops as 16x16x32 instead of 32x32x16 MFMAs is worth +5-7 % at the board's power cap (4 accumulator registers
    written per 16 matrix cycles instead of 16 per 32)."""

    def __init__(self, nbj=8):
        self.NB = 8                             # 16-row blocks per wave tile (128 rows)
        self.NBJ = nbj                          # 16-column blocks per wave tile: 8 (256-wide workgroup tile) or 4 (128-wide)
        self.BN = 32 * nbj
        self.NA, self.NW = 8, nbj
        self.DMA_STRIDE = 4096
        self.A_STAGE, self.W_STAGE = 32768, 4096 * nbj
        self.W_BASE = 65536
        self.SMEM = self.W_BASE + 2 * self.W_STAGE
        self.NACC = 32 * nbj
        self.NFRAG = 8 + nbj                    # activation + weight fragments per sub-step
        self.FW = 4
        self.V0 = 96                            # asm-owned VGPRs: v96.. (two fragment sets, then the address registers)
        self.VN = 2 * self.NFRAG * self.FW
        self.tag = "x%d" % self.BN

    def frag(self, fset, idx):
        return self.V0 + (fset * self.NFRAG + idx) * self.FW


def gen_x4(c, spread=4, rd_per=1):
    """gen_w4's persistent-workgroup K loop (same operands, same LDS-DMA side, same cross-tile prefetch and bias-initialised
    accumulators) for Cfg4x.  Per K step and wave: 128 MFMAs of 16 cycles in two blocks of 64 -- the trailing sub-step (k 32..63
    of the previous K step, fragment set 1) and sub-step 0 (set 0) --, 16 fragment reads per block two per shadow at its head,
    the 16 LDS-DMA pieces one per `spread` shadows of the trailing block (the time spacing of gen_w4's one per two 32-cycle
    shadows)."""
    L = []
    e = lambda t: L.append("  " + t)
    lab = lambda n: L.append(".L%s_%s_%%=:" % (c.tag, n))
    ref = lambda n: ".L%s_%s_%%=" % (c.tag, n)
    VX = c.V0 + c.VN
    fa = {("A", 0): WOP["faA0"], ("W", 0): WOP["faW0"], ("A", 1): vr(VX), ("W", 1): vr(VX + 1)}
    NB = c.NB
    NM = NB * NB

    def reads(stage, s32, fset):
        out = []
        for i in range(NB):
            out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, i), 4), fa[("A", s32)], stage * c.A_STAGE + i * 2048))
        for j in range(NB):
            out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, NB + j), 4), fa[("W", s32)], stage * c.W_STAGE + j * 2048))
        return out

    def mfmas(fset):
        # swapped operands: first = weight fragment (D rows = 16 output channels), second = activation fragment (D columns =
        # 16 output rows): a lane owns output row l % 16 and channels 4 (l / 16) .. + 3 of every 16 x 16 tile
        out = []
        for j in range(NB):
            for i in range(NB):
                acc = ar((j * NB + i) * 4, 4)
                out.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (acc, vr(c.frag(fset, NB + j), 4), vr(c.frag(fset, i), 4), acc))
        return out

    def dma(stage, aregs=None, wregs=None):
        aregs = aregs or [WOP["aoff%d" % i] for i in range(c.NA)]
        wregs = wregs or [WOP["woff%d" % i] for i in range(c.NW)]
        out = []
        for i in range(c.NA):
            out.append(("s_add_u32 m0, s%d, %d" % (S_ADST, stage * c.A_STAGE + i * c.DMA_STRIDE),
                        "global_load_lds_dwordx4 %s, s[%d:%d]" % (aregs[i], S_AB, S_AB + 1)))
        for i in range(c.NW):
            out.append(("s_add_u32 m0, s%d, %d" % (S_WDST, stage * c.W_STAGE + i * c.DMA_STRIDE),
                        "global_load_lds_dwordx4 %s, s[%d:%d]" % (wregs[i], S_WB, S_WB + 1)))
        return out

    def advance():
        return ["s_add_u32 s%d, s%d, 1" % (S_TMP, S_KL),
                "s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NK),
                "s_cselect_b32 s%d, 128, 0" % S_STEP,
                "s_cselect_b32 s%d, s%d, s%d" % (S_KL, S_TMP, S_KL),
                "s_add_u32 s%d, s%d, s%d" % (S_AB, S_AB, S_STEP),
                "s_addc_u32 s%d, s%d, 0" % (S_AB + 1, S_AB + 1),
                "s_add_u32 s%d, s%d, s%d" % (S_WB, S_WB, S_STEP),
                "s_addc_u32 s%d, s%d, 0" % (S_WB + 1, S_WB + 1)]

    def plain(pieces):
        for m0w, d in pieces:
            e(m0w); e("s_nop 0"); e(d)

    # ---- setup
    e("s_mov_b64 s[%d:%d], %s" % (S_AB, S_AB + 1, WOP["abase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_WB, S_WB + 1, WOP["wbase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_PBIAS, S_PBIAS + 1, WOP["bias"]))
    e("s_mov_b32 s%d, %s" % (S_NK, WOP["nk"]))
    e("s_mov_b32 s%d, %s" % (S_ADST, WOP["adst"]))
    e("s_mov_b32 s%d, %s" % (S_WDST, WOP["wdst"]))
    e("s_mov_b32 s%d, %s" % (S_PFLAGS, WOP["flags"]))
    e("s_mov_b32 s%d, %s" % (S_DA, WOP["dA"]))
    e("s_mov_b32 s%d, %s" % (S_DW, WOP["dW"]))
    e("s_mov_b32 s%d, 0" % S_T)
    e("s_mov_b32 s%d, 0" % S_KL)
    e("v_xor_b32_e32 %s, 64, %s" % (fa[("A", 1)], WOP["faA0"]))     # k 32..63: 16-byte chunk index + 4 = ^ 4 under the swizzle
    e("v_xor_b32_e32 %s, 64, %s" % (fa[("W", 1)], WOP["faW0"]))
    e("s_bitcmp1_b32 s%d, %d" % (S_PFLAGS, PF_PREFETCHED))
    e("s_cbranch_scc1 %s" % ref("have0"))
    plain(dma(0))
    lab("have0")
    for a in advance():
        e(a)
    e("s_bitcmp1_b32 s%d, %d" % (S_PFLAGS, PF_BIAS))
    e("s_cbranch_scc0 %s" % ref("nobias"))
    for j in range(NB):   # bias of this lane's 4 channels of column block j -> the idle fragment registers
        e("global_load_dwordx4 %s, %s, s[%d:%d] offset:%d" % (vr(c.V0 + j * 4, 4), WOP["boff"], S_PBIAS, S_PBIAS + 1, j * 64))
    lab("nobias")
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    e("s_bitcmp1_b32 s%d, %d" % (S_PFLAGS, PF_PREFETCHED))
    e("s_cbranch_scc1 %s" % ref("have1"))
    plain(dma(1))
    lab("have1")
    assert spread * (c.NA + c.NW - 1) < NM, "all pieces are issued in the trailing block: entry0 re-issues none"
    for a in advance():
        e(a)
    e("s_bitcmp1_b32 s%d, %d" % (S_PFLAGS, PF_BIAS))
    e("s_cbranch_scc0 %s" % ref("zero"))
    for j in range(NB):
        for i in range(NB):
            for r in range(4):
                e("v_accvgpr_write_b32 %s, %s" % (ar((j * NB + i) * 4 + r), vr(c.V0 + j * 4 + r)))
    e("s_branch %s" % ref("inited"))
    lab("zero")
    for r in range(c.NACC):
        e("v_accvgpr_write_b32 %s, 0" % ar(r))
    lab("inited")
    e("s_nop 1")
    for r in reads(0, 0, 0):
        e(r)
    e("s_branch %s" % ref("entry0"))

    def step(cur):
        pieces = dma(cur ^ 1)
        slots = [[] for _ in range(2 * NM)]
        dma_slots = [spread * j for j in range(len(pieces))]
        e(pieces[0][0])
        for j, (m0w, d) in enumerate(pieces):
            slots[dma_slots[j]].append(d)
            if j + 1 < len(pieces):
                slots[dma_slots[j]].append(pieces[j + 1][0])
        last = dma_slots[-1]
        for blk, (s32, fset) in enumerate(((0, 0), (1, 1))):
            rd = reads(cur, s32, fset)
            free = [i for i in range(blk * NM, (blk + 1) * NM) if i not in dma_slots]
            for i in range((len(rd) + rd_per - 1) // rd_per):      # a ds_read_b128 costs ~9 cycles of a 16-cycle shadow
                slots[free[i]] += rd[rd_per * i: rd_per * i + rd_per]
        for blk in range(2):
            if blk == 1:
                lab("entry%d" % cur)
                e("s_waitcnt lgkmcnt(0)")
            mf = mfmas(1 if blk == 0 else 0)
            for i, m in enumerate(mf):
                e(m)
                for x in slots[blk * NM + i]:
                    e(x)
                if blk * NM + i == last:
                    for a in advance():
                        e(a)

    for k in range(2):
        lab("step%d" % k)
        step(k)
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e("s_barrier")
        e("s_add_u32 s%d, s%d, 1" % (S_T, S_T))
        e("s_cmp_lt_u32 s%d, s%d" % (S_T, S_NK))
        e("s_cbranch_scc0 %s" % ref("exit"))
        if k == 1:
            e("s_branch %s" % ref("step0"))
    lab("exit")
    e("s_bitcmp1_b32 s%d, %d" % (S_PFLAGS, PF_HAS_NEXT))
    e("s_cbranch_scc0 %s" % ref("last"))
    e("s_mov_b64 s[%d:%d], %s" % (S_AB, S_AB + 1, WOP["abase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_WB, S_WB + 1, WOP["wbase"]))
    # next tile's per-lane source offsets into fragment set 0 (its last MFMAs were issued before the barrier)
    an = [vr(c.V0 + i) for i in range(c.NA)]
    wn = [vr(c.V0 + c.NA + i) for i in range(c.NW)]
    for i in range(c.NA):
        e("v_add_u32_e32 %s, s%d, %s" % (an[i], S_DA, WOP["aoff%d" % i]))
    for i in range(c.NW):
        e("v_add_u32_e32 %s, s%d, %s" % (wn[i], S_DW, WOP["woff%d" % i]))
    mf = mfmas(1)
    pieces = dma(0, an, wn)
    e(pieces[0][0])
    for i, m in enumerate(mf):
        e(m)
        if i % spread == 0 and i // spread < len(pieces):
            k = i // spread
            e(pieces[k][1])
            if k + 1 < len(pieces):
                e(pieces[k + 1][0])
    e("s_cmp_gt_u32 s%d, 1" % S_NK)
    e("s_cselect_b32 s%d, 128, 0" % S_STEP)
    e("s_add_u32 s%d, s%d, s%d" % (S_AB, S_AB, S_STEP))
    e("s_addc_u32 s%d, s%d, 0" % (S_AB + 1, S_AB + 1))
    e("s_add_u32 s%d, s%d, s%d" % (S_WB, S_WB, S_STEP))
    e("s_addc_u32 s%d, s%d, 0" % (S_WB + 1, S_WB + 1))
    plain(dma(1, an, wn))
    e("s_branch %s" % ref("end"))
    lab("last")
    for m in mfmas(1):
        e(m)
    lab("end")
    e("s_nop 15")
    e("s_nop 15")
    return L


def gen_fp8(c):
    """K loop of the fp8 (OCP e4m3) GEMM: the SAME tile, LDS image (128-byte rows = 128 K elements now), LDS-DMA
    loaders and operands as gen(); one v_mfma_f32_32x32x64_f8f6f4 (64 cycles, 2x the bf16 MAC rate) consumes what
    two consecutive bf16 k-sub-steps read: a 32-byte fragment = the 16-byte chunks (2 ks | hi) of k-sub-steps
    ks = 2 kp and 2 kp + 1.  (Which K element lands in which byte of the MFMA operand does not matter: activation
    and weight fragments are assembled identically, and a contraction is invariant under a common K permutation.)
    One K step of a wave = 2 k-sub-steps: reads of sub-step 0 | the trailing sub-step 1 of the previous K step with
    the LDS-DMA of the other stage in its shadows | sub-step 0 with the reads of sub-step 1 in its shadows |
    vmcnt(0) lgkmcnt(0), barrier."""
    assert c.FP8
    L = []
    e = lambda t: L.append("  " + t)
    lab = lambda n: L.append(".L%s_%s_%%=:" % (c.tag, n))

    def reads(stage, kp, fset):
        out = []
        for half in range(2):
            ks = 2 * kp + half
            for tm in range(c.TM):
                out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, tm) + 4 * half, 4), OP["faA%d" % ks], stage * c.A_STAGE + tm * 4096))
            for tn in range(c.TN):
                out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, c.TM + tn) + 4 * half, 4), OP["faW%d" % ks], stage * c.W_STAGE + tn * 4096))
        return out

    def mfmas(fset):
        out = []
        for tn in range(c.TN):
            for tm in range(c.TM):
                acc = ar((tn * c.TM + tm) * 16, 16)
                out.append("v_mfma_f32_32x32x64_f8f6f4 %s, %s, %s, %s" % (acc, vr(c.frag(fset, c.TM + tn), 8), vr(c.frag(fset, tm), 8), acc))
        return out

    def dma(stage):
        out = []
        for i in range(c.NA):
            out.append(("s_add_u32 m0, s%d, %d" % (S_ADST, stage * c.A_STAGE + i * 8192),
                        "global_load_lds_dwordx4 %s, s[%d:%d]" % (OP["aoff%d" % i], S_AB, S_AB + 1)))
        for i in range(c.NW):
            out.append(("s_add_u32 m0, s%d, %d" % (S_WDST, stage * c.W_STAGE + i * 8192),
                        "global_load_lds_dwordx4 %s, s[%d:%d]" % (OP["woff%d" % i], S_WB, S_WB + 1)))
        return out

    def advance():
        return ["s_add_u32 s%d, s%d, 1" % (S_TMP, S_KL),
                "s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NK),
                "s_cselect_b32 s%d, 128, 0" % S_STEP,
                "s_cselect_b32 s%d, s%d, s%d" % (S_KL, S_TMP, S_KL),
                "s_add_u32 s%d, s%d, s%d" % (S_AB, S_AB, S_STEP),
                "s_addc_u32 s%d, s%d, 0" % (S_AB + 1, S_AB + 1),
                "s_add_u32 s%d, s%d, s%d" % (S_WB, S_WB, S_STEP),
                "s_addc_u32 s%d, s%d, 0" % (S_WB + 1, S_WB + 1)]

    def spread(mf, fill_items):
        """MFMAs with the filler instructions spread evenly over their shadows"""
        n = len(mf)
        per = [[] for _ in mf]
        for i, it in enumerate(fill_items):
            per[i * n // max(1, len(fill_items))].append(it)
        for m, f in zip(mf, per):
            e(m)
            for x in f:
                e(x)

    e("s_mov_b64 s[%d:%d], %s" % (S_AB, S_AB + 1, OP["abase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_WB, S_WB + 1, OP["wbase"]))
    e("s_mov_b32 s%d, %s" % (S_NK, OP["nk"]))
    e("s_mov_b32 s%d, %s" % (S_ADST, OP["adst"]))
    e("s_mov_b32 s%d, %s" % (S_WDST, OP["wdst"]))
    e("s_mov_b32 s%d, 0" % S_T)
    e("s_mov_b32 s%d, 0" % S_KL)
    for r in range(c.NACC):
        e("v_accvgpr_write_b32 %s, 0" % ar(r))
    for m0w, d in dma(0):
        e(m0w); e("s_nop 0"); e(d)
    for a in advance():
        e(a)
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    for m0w, d in dma(1):
        e(m0w); e("s_nop 0"); e(d)
    for a in advance():
        e(a)
    for r in reads(0, 0, 0):
        e(r)
    e("s_branch .L%s_entry0_%%=" % c.tag)

    for k in range(2):
        cur = k
        lab("step%d" % k)
        for r in reads(cur, 0, 0):
            e(r)
        mf = mfmas(1)                      # trailing sub-step 1 of the previous K step
        pieces = dma(cur ^ 1)
        e(pieces[0][0])                    # M0 write before MFMA i, its DMA after it, the next M0 write behind that
        for i, m in enumerate(mf):
            e(m)
            if i < len(pieces):
                e(pieces[i][1])
                if i + 1 < len(pieces):
                    e(pieces[i + 1][0])
        for m0w, d in pieces[len(mf):]:
            e(m0w); e("s_nop 0"); e(d)
        for a in advance():
            e(a)
        lab("entry%d" % k)
        e("s_waitcnt lgkmcnt(0)")
        spread(mfmas(0), reads(cur, 1, 1))
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e("s_barrier")
        e("s_add_u32 s%d, s%d, 1" % (S_T, S_T))
        e("s_cmp_lt_u32 s%d, s%d" % (S_T, S_NK))
        e("s_cbranch_scc0 .L%s_exit_%%=" % c.tag)
        if k == 1:
            e("s_branch .L%s_step0_%%=" % c.tag)
    lab("exit")
    for m in mfmas(1):
        e(m)
    e("s_nop 15")
    e("s_nop 15")
    return L


SEG_OPERANDS = ["faA0", "faA1", "faA2", "faA3", "faW0", "faW1", "faW2", "faW3", "aoffc0", "aoffc1", "aoffc2", "aoffc3",
                "aoffn0", "aoffn1", "aoffn2", "aoffn3", "woff0", "woff1", "woff2", "woff3",
                "xbase", "wbase", "nk", "adst", "wdst", "flags"]
SOP = {n: "%%%d" % i for i, n in enumerate(SEG_OPERANDS)}
S_XB, S_FLAGS, S_AE, S_WE = 52, 54, 56, 58  # x base (pair), flags, A / W base of the step being fetched (pairs)
SEG_S_LAST = 59


def gen_segment(c):
    """One K SEGMENT of the implicit-GEMM convolution (csrc/conv3d_256.hip): nk (even) K steps that share the per-lane
    activation offsets aoffc (one filter tap: the A rows are gathered voxels), accumulating into the AGPRs that persist
    between calls.  The LDS-DMA issued in the last K step already fetches the NEXT segment's first step (offsets
    aoffn), so a call never waits for HBM except the very first one (flags bit 0: zero the accumulators, load stage 0);
    flags bit 1 = last segment (the weight loader must not run past the end of the K axis).  The weight operand is
    contiguous along K across segments: its scalar base just keeps advancing."""
    L = []
    e = lambda t: L.append("  " + t)
    lab = lambda n: L.append(".Ls%s_%s_%%=:" % (c.tag, n))
    VT = c.V0 + c.VN                       # 4 temporaries: the activation offsets of the step being fetched

    def reads(stage, ks, fset):
        out = []
        for tm in range(c.TM):
            out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, tm), 4), SOP["faA%d" % ks], stage * c.A_STAGE + tm * 4096))
        for tn in range(c.TN):
            out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, c.TM + tn), 4), SOP["faW%d" % ks], stage * c.W_STAGE + tn * 4096))
        return out

    def mfmas(fset):
        out = []
        for tn in range(c.TN):
            for tm in range(c.TM):
                acc = ar((tn * c.TM + tm) * 16, 16)
                out.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, vr(c.frag(fset, c.TM + tn), 4), vr(c.frag(fset, tm), 4), acc))
        return out

    def dma(stage, aoff_regs):
        out = []
        for i in range(c.NA):
            out.append(("s_add_u32 m0, s%d, %d" % (S_ADST, stage * c.A_STAGE + i * 8192),
                        "global_load_lds_dwordx4 %s, s[%d:%d]" % (aoff_regs[i], S_AE, S_AE + 1)))
        for i in range(c.NW):
            out.append(("s_add_u32 m0, s%d, %d" % (S_WDST, stage * c.W_STAGE + i * 8192),
                        "global_load_lds_dwordx4 %s, s[%d:%d]" % (SOP["woff%d" % i], S_WE, S_WE + 1)))
        return out

    # ---- setup
    e("s_mov_b64 s[%d:%d], %s" % (S_XB, S_XB + 1, SOP["xbase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_WB, S_WB + 1, SOP["wbase"]))
    e("s_mov_b32 s%d, %s" % (S_NK, SOP["nk"]))
    e("s_mov_b32 s%d, %s" % (S_ADST, SOP["adst"]))
    e("s_mov_b32 s%d, %s" % (S_WDST, SOP["wdst"]))
    e("s_mov_b32 s%d, %s" % (S_FLAGS, SOP["flags"]))
    e("s_mov_b32 s%d, 0" % S_T)
    e("s_bitcmp1_b32 s%d, 0" % S_FLAGS)
    e("s_cbranch_scc0 .Ls%s_go_%%=" % c.tag)
    # first segment of a tile: clear the accumulators, fetch step 0 of this segment into stage 0
    for r in range(c.NACC):
        e("v_accvgpr_write_b32 %s, 0" % ar(r))
    e("s_mov_b64 s[%d:%d], s[%d:%d]" % (S_AE, S_AE + 1, S_XB, S_XB + 1))
    e("s_mov_b64 s[%d:%d], s[%d:%d]" % (S_WE, S_WE + 1, S_WB, S_WB + 1))
    for m0w, d in dma(0, [SOP["aoffc%d" % i] for i in range(c.NA)]):
        e(m0w); e("s_nop 0"); e(d)
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    lab("go")

    for k in range(2):
        cur = k
        lab("step%d" % k)
        for r in reads(cur, 0, 0):
            e(r)
        # the step to fetch: t+1 inside the segment (same tap: offsets aoffc, base x + 128 (t+1)) or step 0 of the next
        # segment (offsets aoffn, base x)
        e("s_add_u32 s%d, s%d, 1" % (S_TMP, S_T))
        e("s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NK))           # scc = the step to fetch lies inside this segment
        e("s_cselect_b64 vcc, -1, 0")
        e("s_cselect_b32 s%d, s%d, 0" % (S_STEP, S_TMP))      # A: step t+1 of this tap, or step 0 of the next tap
        e("s_cselect_b32 s%d, 0, s%d" % (S_KL, S_FLAGS))      # flags only at the segment's last step
        e("s_lshl_b32 s%d, s%d, 7" % (S_STEP, S_STEP))
        e("s_add_u32 s%d, s%d, s%d" % (S_AE, S_XB, S_STEP))
        e("s_addc_u32 s%d, s%d, 0" % (S_AE + 1, S_XB + 1))
        e("s_bitcmp1_b32 s%d, 1" % S_KL)                      # scc = last step of the LAST segment: K axis ends here
        e("s_cselect_b32 s%d, s%d, s%d" % (S_STEP, S_T, S_TMP))  # W: contiguous along K across taps; re-fetch at the end
        e("s_lshl_b32 s%d, s%d, 7" % (S_STEP, S_STEP))
        e("s_add_u32 s%d, s%d, s%d" % (S_WE, S_WB, S_STEP))
        e("s_addc_u32 s%d, s%d, 0" % (S_WE + 1, S_WB + 1))
        for i in range(c.NA):
            e("v_cndmask_b32_e32 %s, %s, %s, vcc" % (vr(VT + i), SOP["aoffn%d" % i], SOP["aoffc%d" % i]))
        pieces = dma(cur ^ 1, [vr(VT + i) for i in range(c.NA)])
        e("s_waitcnt lgkmcnt(0)")
        mf = mfmas(0)
        rd = reads(cur, 1, 1)
        e(pieces[0][0])
        for i, m in enumerate(mf):
            e(m)
            if i < len(rd):
                e(rd[i])
            if i < len(pieces):
                e(pieces[i][1])
                if i + 1 < len(pieces):
                    e(pieces[i + 1][0])
        for m0w, d in pieces[len(mf):]:
            e(m0w); e("s_nop 0"); e(d)
        for ks in range(1, 4):
            fset = ks % 2
            e("s_waitcnt lgkmcnt(0)")
            mf = mfmas(fset)
            rd = reads(cur, ks + 1, fset ^ 1) if ks < 3 else []
            for i, m in enumerate(mf):
                e(m)
                if i < len(rd):
                    e(rd[i])
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e("s_barrier")
        e("s_add_u32 s%d, s%d, 1" % (S_T, S_T))
        if k == 1:
            e("s_cmp_lt_u32 s%d, s%d" % (S_T, S_NK))
            e("s_cbranch_scc1 .Ls%s_step0_%%=" % c.tag)
    e("s_nop 15")
    e("s_nop 15")
    return L


CONV_OPERANDS = ["faA0", "faA1", "faA2", "faA3", "faW0", "faW1", "faW2", "faW3", "arow0", "arow1", "arow2", "arow3",
                 "chk0", "chk1", "chk2", "chk3", "woff0", "woff1", "woff2", "woff3",
                 "xbase", "wbase", "nk", "nkt", "adst", "wdst"]
COP = {n: "%%%d" % i for i, n in enumerate(CONV_OPERANDS)}
S_NKT, S_SIT, S_TAPOFF, S_XB2 = 52, 53, 54, 56     # K steps per tap, step inside the tap, tap * 1024, x base (pair)
CONV_S_LAST = 57


def gen_conv(c):
    """The whole K axis of the implicit-GEMM convolution (csrc/conv3d_256.hip) in ONE call: the gemm256 pipeline of
    gen() with gathered A rows.  K = taps x (Cin / 64) steps; inside a tap a step only advances the channel block
    (scalar base + 128 B), at a tap boundary every row slot takes a new voxel offset.  Those offsets sit in an LDS
    table [tap][256 rows] (4 B each, built by the wrapper before the call; 27 KB next to the two 64 KB stages): each
    K step re-reads its 4 entries for the step it is about to fetch (4 ds_read_b32, no branch) and adds the lane's
    swizzled channel-chunk offset."""
    L = []
    e = lambda t: L.append("  " + t)
    lab = lambda n: L.append(".Lc%s_%s_%%=:" % (c.tag, n))
    VR = c.V0 + c.VN          # 4 raw table entries (ds_read destinations)
    VO = VR + 4               # 4 offsets of the step being fetched
    VA = VO + 4               # 4 table addresses

    def reads(stage, ks, fset):
        out = []
        for tm in range(c.TM):
            out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, tm), 4), COP["faA%d" % ks], stage * c.A_STAGE + tm * 4096))
        for tn in range(c.TN):
            out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, c.TM + tn), 4), COP["faW%d" % ks], stage * c.W_STAGE + tn * 4096))
        return out

    def mfmas(fset):
        out = []
        for tn in range(c.TN):
            for tm in range(c.TM):
                acc = ar((tn * c.TM + tm) * 16, 16)
                out.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, vr(c.frag(fset, c.TM + tn), 4), vr(c.frag(fset, tm), 4), acc))
        return out

    def dma(stage):
        out = []
        for i in range(c.NA):
            out.append(("s_add_u32 m0, s%d, %d" % (S_ADST, stage * c.A_STAGE + i * 8192),
                        "global_load_lds_dwordx4 %s, s[%d:%d]" % (vr(VO + i), S_AB, S_AB + 1)))
        for i in range(c.NW):
            out.append(("s_add_u32 m0, s%d, %d" % (S_WDST, stage * c.W_STAGE + i * 8192),
                        "global_load_lds_dwordx4 %s, s[%d:%d]" % (COP["woff%d" % i], S_WB, S_WB + 1)))
        return out

    def table_reads():
        """raw offsets of the step the loaders point at (tap = S_TAPOFF / 1024)"""
        out = []
        for i in range(c.NA):
            out.append("v_add_u32_e32 %s, s%d, %s" % (vr(VA + i), S_TAPOFF, COP["arow%d" % i]))
        for i in range(c.NA):
            out.append("ds_read_b32 %s, %s" % (vr(VR + i), vr(VA + i)))
        return out

    def table_adds():
        return ["v_add_u32_e32 %s, %s, %s" % (vr(VO + i), vr(VR + i), COP["chk%d" % i]) for i in range(c.NA)]

    def advance():
        """point the loaders at the next K step (tap-major): inside a tap the A base moves one channel block, at a
        tap boundary it returns to x and the table row changes; W is contiguous along K.  Past the last step: stay."""
        return ["s_add_u32 s%d, s%d, 1" % (S_TMP, S_KL),
                "s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NK),
                "s_cselect_b32 s%d, 128, 0" % S_STEP,                       # W step (0 at the very end: re-fetch)
                "s_cselect_b32 s%d, s%d, s%d" % (S_KL, S_TMP, S_KL),
                "s_cselect_b32 s%d, 1, 0" % S_TMP,                           # 1 while advancing
                "s_add_u32 s%d, s%d, s%d" % (S_WB, S_WB, S_STEP),
                "s_addc_u32 s%d, s%d, 0" % (S_WB + 1, S_WB + 1),
                "s_add_u32 s%d, s%d, s%d" % (S_SIT, S_SIT, S_TMP),           # step inside the tap
                "s_cmp_ge_u32 s%d, s%d" % (S_SIT, S_NKT),                     # tap boundary?
                "s_cselect_b32 s%d, 0, s%d" % (S_SIT, S_SIT),
                "s_cselect_b32 s%d, 1024, 0" % S_STEP,
                "s_add_u32 s%d, s%d, s%d" % (S_TAPOFF, S_TAPOFF, S_STEP),
                "s_lshl_b32 s%d, s%d, 7" % (S_STEP, S_SIT),
                "s_add_u32 s%d, s%d, s%d" % (S_AB, S_XB2, S_STEP),
                "s_addc_u32 s%d, s%d, 0" % (S_AB + 1, S_XB2 + 1)]

    # ---- setup
    e("s_mov_b64 s[%d:%d], %s" % (S_XB2, S_XB2 + 1, COP["xbase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_AB, S_AB + 1, COP["xbase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_WB, S_WB + 1, COP["wbase"]))
    e("s_mov_b32 s%d, %s" % (S_NK, COP["nk"]))
    e("s_mov_b32 s%d, %s" % (S_NKT, COP["nkt"]))
    e("s_mov_b32 s%d, %s" % (S_ADST, COP["adst"]))
    e("s_mov_b32 s%d, %s" % (S_WDST, COP["wdst"]))
    for sreg in (S_T, S_KL, S_SIT, S_TAPOFF):
        e("s_mov_b32 s%d, 0" % sreg)
    for r in range(c.NACC):
        e("v_accvgpr_write_b32 %s, 0" % ar(r))
    # ---- prologue: offsets of step 0, stage 0; then what step 0 does before its entry point
    for t in table_reads():
        e(t)
    e("s_waitcnt lgkmcnt(0)")
    for t in table_adds():
        e(t)
    e("s_nop 1")
    for m0w, d in dma(0):
        e(m0w); e("s_nop 0"); e(d)
    for a in advance():
        e(a)
    for t in table_reads():
        e(t)
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")     # stage 0 landed AND its DMA has consumed the offset registers
    for t in table_adds():
        e(t)
    e("s_barrier")
    e("s_nop 1")
    for m0w, d in dma(1):
        e(m0w); e("s_nop 0"); e(d)
    for a in advance():
        e(a)
    for r in reads(0, 0, 0):
        e(r)
    for t in table_reads():
        e(t)
    e("s_branch .Lc%s_entry0_%%=" % c.tag)

    for k in range(2):
        cur = k
        lab("step%d" % k)
        for r in reads(cur, 0, 0):
            e(r)
        mf = mfmas(1)                      # trailing k-sub-step 3 of the previous K step
        pieces = dma(cur ^ 1)              # fetch of the step the loaders point at (offsets VO: refreshed last step)
        e(pieces[0][0])
        for i, m in enumerate(mf):
            e(m)
            if i < len(pieces):
                e(pieces[i][1])
                if i + 1 < len(pieces):
                    e(pieces[i + 1][0])
        for m0w, d in pieces[len(mf):]:
            e(m0w); e("s_nop 0"); e(d)
        for a in advance():
            e(a)
        for t in table_reads():            # offsets for the NEXT fetch; they join the fragment reads' wait below
            e(t)
        lab("entry%d" % k)
        for ks in range(3):
            fset = ks % 2
            e("s_waitcnt lgkmcnt(0)")
            mf = mfmas(fset)
            rd = reads(cur, ks + 1, fset ^ 1)
            for i, m in enumerate(mf):
                e(m)
                if i < len(rd):
                    e(rd[i])
            if ks == 2:
                # VO <- raw + chunk: only now, when this step's LDS-DMA instructions (issued above, reading VO) are
                # guaranteed to have read their address registers (they precede >= 16 MFMAs in program order)
                for t in table_adds():
                    e(t)
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e("s_barrier")
        e("s_add_u32 s%d, s%d, 1" % (S_T, S_T))
        e("s_cmp_lt_u32 s%d, s%d" % (S_T, S_NK))
        e("s_cbranch_scc0 .Lc%s_exit_%%=" % c.tag)
        if k == 1:
            e("s_branch .Lc%s_step0_%%=" % c.tag)
    lab("exit")
    for m in mfmas(1):
        e(m)
    e("s_nop 15")
    e("s_nop 15")
    return L

CW_OPERANDS = ["faA0", "faW0", "arow0", "chk"] + ["woff%d" % i for i in range(8)] + ["xbase", "wbase", "nk", "nkt", "adst", "wdst"]
CWOP = {n: "%%%d" % i for i, n in enumerate(CW_OPERANDS)}


def gen_conv_w4(c, spread=2):
    """gen_conv on the 4-wave layout of Cfg4 with gen_w4's spread schedule: the whole K axis (all filter taps) of the
    implicit-GEMM convolution in one call, wave tile 128 x 128, one LDS-DMA piece per `spread` MFMA shadows.  A wave's 8
    activation pieces cover tile rows 8 (wave + 4 i) + lane / 8: their voxel offsets sit 128 bytes apart in the LDS table
    row of the tap (one address register + immediate offsets), and the swizzled channel-chunk offset is the same for all
    8 (the row's swizzle key (r >> 1) & 7 does not depend on i).  Per K step: pieces A0..A7 in the trailing sub-step,
    W0..W7 in sub-step 0, then the loaders advance, sub-step 1 re-reads the table for the step they now point at and
    sub-step 2 adds the chunk offset (by then this step's pieces have long read their address registers)."""
    L = []
    e = lambda t: L.append("  " + t)
    lab = lambda n: L.append(".Lcw_%s_%%=:" % n)
    ref = lambda n: ".Lcw_%s_%%=" % n
    VX = c.V0 + c.VN
    fa = {("A", 0): CWOP["faA0"], ("W", 0): CWOP["faW0"]}
    for ks in range(1, 4):
        fa[("A", ks)] = vr(VX + ks - 1)
        fa[("W", ks)] = vr(VX + 3 + ks - 1)
    VA = VX + 6                # table address of the tap being fetched
    VR = VA + 1                # 8 raw table entries
    VO = VR + c.NA             # 8 offsets of the step being fetched
    NM = c.TM * c.TN

    def reads(stage, ks, fset):
        out = []
        for tm in range(c.TM):
            out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, tm), 4), fa[("A", ks)], stage * c.A_STAGE + tm * 4096))
        for tn in range(c.TN):
            out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, c.TM + tn), 4), fa[("W", ks)], stage * c.W_STAGE + tn * 4096))
        return out

    def mfmas(fset):
        out = []
        for tn in range(c.TN):
            for tm in range(c.TM):
                acc = ar((tn * c.TM + tm) * 16, 16)
                out.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, vr(c.frag(fset, c.TM + tn), 4), vr(c.frag(fset, tm), 4), acc))
        return out

    def dma(stage):
        out = []
        for i in range(c.NA):
            out.append(("s_add_u32 m0, s%d, %d" % (S_ADST, stage * c.A_STAGE + i * c.DMA_STRIDE),
                        "global_load_lds_dwordx4 %s, s[%d:%d]" % (vr(VO + i), S_AB, S_AB + 1)))
        for i in range(c.NW):
            out.append(("s_add_u32 m0, s%d, %d" % (S_WDST, stage * c.W_STAGE + i * c.DMA_STRIDE),
                        "global_load_lds_dwordx4 %s, s[%d:%d]" % (CWOP["woff%d" % i], S_WB, S_WB + 1)))
        return out

    def table_reads():
        return ["v_add_u32_e32 %s, s%d, %s" % (vr(VA), S_TAPOFF, CWOP["arow0"])] + \
               ["ds_read_b32 %s, %s offset:%d" % (vr(VR + i), vr(VA), i * 128) for i in range(c.NA)]

    def table_adds():
        return ["v_add_u32_e32 %s, %s, %s" % (vr(VO + i), vr(VR + i), CWOP["chk"]) for i in range(c.NA)]

    def advance():     # gen_conv's: tap-major K axis, clamp at the end
        return ["s_add_u32 s%d, s%d, 1" % (S_TMP, S_KL),
                "s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NK),
                "s_cselect_b32 s%d, 128, 0" % S_STEP,
                "s_cselect_b32 s%d, s%d, s%d" % (S_KL, S_TMP, S_KL),
                "s_cselect_b32 s%d, 1, 0" % S_TMP,
                "s_add_u32 s%d, s%d, s%d" % (S_WB, S_WB, S_STEP),
                "s_addc_u32 s%d, s%d, 0" % (S_WB + 1, S_WB + 1),
                "s_add_u32 s%d, s%d, s%d" % (S_SIT, S_SIT, S_TMP),
                "s_cmp_ge_u32 s%d, s%d" % (S_SIT, S_NKT),
                "s_cselect_b32 s%d, 0, s%d" % (S_SIT, S_SIT),
                "s_cselect_b32 s%d, 1024, 0" % S_STEP,
                "s_add_u32 s%d, s%d, s%d" % (S_TAPOFF, S_TAPOFF, S_STEP),
                "s_lshl_b32 s%d, s%d, 7" % (S_STEP, S_SIT),
                "s_add_u32 s%d, s%d, s%d" % (S_AB, S_XB2, S_STEP),
                "s_addc_u32 s%d, s%d, 0" % (S_AB + 1, S_XB2 + 1)]

    def plain(pieces):
        for m0w, d in pieces:
            e(m0w); e("s_nop 0"); e(d)

    # ---- setup
    e("s_mov_b64 s[%d:%d], %s" % (S_XB2, S_XB2 + 1, CWOP["xbase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_AB, S_AB + 1, CWOP["xbase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_WB, S_WB + 1, CWOP["wbase"]))
    e("s_mov_b32 s%d, %s" % (S_NK, CWOP["nk"]))
    e("s_mov_b32 s%d, %s" % (S_NKT, CWOP["nkt"]))
    e("s_mov_b32 s%d, %s" % (S_ADST, CWOP["adst"]))
    e("s_mov_b32 s%d, %s" % (S_WDST, CWOP["wdst"]))
    for sreg in (S_T, S_KL, S_SIT, S_TAPOFF):
        e("s_mov_b32 s%d, 0" % sreg)
    for ks in range(1, 4):
        e("v_xor_b32_e32 %s, %d, %s" % (fa[("A", ks)], ks << 5, CWOP["faA0"]))
        e("v_xor_b32_e32 %s, %d, %s" % (fa[("W", ks)], ks << 5, CWOP["faW0"]))
    for r in range(c.NACC):
        e("v_accvgpr_write_b32 %s, 0" % ar(r))
    for t in table_reads():
        e(t)
    e("s_waitcnt lgkmcnt(0)")
    for t in table_adds():
        e(t)
    e("s_nop 1")
    plain(dma(0))
    for a in advance():
        e(a)
    for t in table_reads():
        e(t)
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")     # stage 0 landed AND its DMA has consumed the offset registers
    for t in table_adds():
        e(t)
    e("s_barrier")
    e("s_nop 1")
    plain(dma(1))                          # entry0 re-issues the W pieces (same step, same data), then advances
    first_entry_piece = (NM + spread - 1) // spread
    assert first_entry_piece >= c.NA, "entry0 must not re-issue activation pieces (their offsets would be stale)"
    e(dma(1)[first_entry_piece][0])
    for r in reads(0, 0, 0):
        e(r)
    e("s_branch %s" % ref("entry0"))

    def step(cur):
        pieces = dma(cur ^ 1)
        slots = [[] for _ in range(4 * NM)]
        dma_slots = [spread * j for j in range(len(pieces))]
        assert dma_slots[-1] < 2 * NM
        e(pieces[0][0])
        for j, (m0w, d) in enumerate(pieces):
            slots[dma_slots[j]].append(d)
            if j + 1 < len(pieces):
                slots[dma_slots[j]].append(pieces[j + 1][0])
        last = dma_slots[-1]
        for blk, (ks, fset) in enumerate(((0, 0), (1, 1), (2, 0), (3, 1))):
            rd = reads(cur, ks, fset)
            free = [i for i in range(blk * NM, (blk + 1) * NM - 3) if i not in dma_slots]
            for i in range((len(rd) + 1) // 2):
                slots[free[i]] += rd[2 * i: 2 * i + 2]
        tr = table_reads()                                     # sub-step 1, behind its fragment reads
        base = 2 * NM + (c.NFRAG + 1) // 2
        slots[base] += tr[:3]
        for i in range(3, len(tr), 2):
            slots[base + 1 + (i - 3) // 2] += tr[i: i + 2]
        ta = table_adds()                                      # sub-step 2, behind its fragment reads
        base = 3 * NM + (c.NFRAG + 1) // 2
        for i, t in enumerate(ta):
            slots[base + i].append(t)
        for blk in range(4):
            if blk == 1:
                lab("entry%d" % cur)
            if blk >= 1:
                e("s_waitcnt lgkmcnt(0)")
            mf = mfmas(1 if blk % 2 == 0 else 0)
            for i, m in enumerate(mf):
                e(m)
                for x in slots[blk * NM + i]:
                    e(x)
                if blk * NM + i == last:
                    for a in advance():
                        e(a)

    for k in range(2):
        lab("step%d" % k)
        step(k)
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e("s_barrier")
        e("s_add_u32 s%d, s%d, 1" % (S_T, S_T))
        e("s_cmp_lt_u32 s%d, s%d" % (S_T, S_NK))
        e("s_cbranch_scc0 %s" % ref("exit"))
        if k == 1:
            e("s_branch %s" % ref("step0"))
    lab("exit")
    for m in mfmas(1):
        e(m)
    e("s_nop 15")
    e("s_nop 15")
    return L

def gen_conv_x4(c, spread=4):
    """gen_conv_w4 (table-driven implicit-GEMM convolution, whole K axis in one call, 4 waves) on Cfg4x's compute side: 128
    v_mfma_f32_16x16x32_bf16 per K step in two blocks of 64 (gen_x4).  Per K step: the 8 activation + 8 weight LDS-DMA pieces one
    per `spread` shadows of the trailing block, then the loaders advance; the other block re-reads the offset table behind its
    fragment reads for the step they now point at, and the chunk offset is added behind the end-of-step wait (this step's
    pieces have long read their address registers)."""
    L = []
    e = lambda t: L.append("  " + t)
    lab = lambda n: L.append(".Lc%s_%s_%%=:" % (c.tag, n))
    ref = lambda n: ".Lc%s_%s_%%=" % (c.tag, n)
    VX = c.V0 + c.VN
    fa = {("A", 0): CWOP["faA0"], ("W", 0): CWOP["faW0"], ("A", 1): vr(VX), ("W", 1): vr(VX + 1)}
    VA = VX + 2                # table address of the tap being fetched
    VR = VA + 1                # 8 raw table entries
    VO = VR + c.NA             # 8 offsets of the step being fetched
    NB, NBJ = c.NB, c.NBJ
    NM = NB * NBJ

    def reads(stage, s32, fset):
        out = []
        for i in range(NB):
            out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, i), 4), fa[("A", s32)], stage * c.A_STAGE + i * 2048))
        for j in range(NBJ):
            out.append("ds_read_b128 %s, %s offset:%d" % (vr(c.frag(fset, NB + j), 4), fa[("W", s32)], stage * c.W_STAGE + j * 2048))
        return out

    def mfmas(fset):
        out = []
        for j in range(NBJ):
            for i in range(NB):
                acc = ar((j * NB + i) * 4, 4)
                out.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (acc, vr(c.frag(fset, NB + j), 4), vr(c.frag(fset, i), 4), acc))
        return out

    def dma(stage):
        out = []
        for i in range(c.NA):
            out.append(("s_add_u32 m0, s%d, %d" % (S_ADST, stage * c.A_STAGE + i * c.DMA_STRIDE),
                        "global_load_lds_dwordx4 %s, s[%d:%d]" % (vr(VO + i), S_AB, S_AB + 1)))
        for i in range(c.NW):
            out.append(("s_add_u32 m0, s%d, %d" % (S_WDST, stage * c.W_STAGE + i * c.DMA_STRIDE),
                        "global_load_lds_dwordx4 %s, s[%d:%d]" % (CWOP["woff%d" % i], S_WB, S_WB + 1)))
        return out

    def table_reads():
        return ["v_add_u32_e32 %s, s%d, %s" % (vr(VA), S_TAPOFF, CWOP["arow0"])] + \
               ["ds_read_b32 %s, %s offset:%d" % (vr(VR + i), vr(VA), i * 128) for i in range(c.NA)]

    def table_adds():
        return ["v_add_u32_e32 %s, %s, %s" % (vr(VO + i), vr(VR + i), CWOP["chk"]) for i in range(c.NA)]

    def advance():     # gen_conv's: tap-major K axis, clamp at the end
        return ["s_add_u32 s%d, s%d, 1" % (S_TMP, S_KL),
                "s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NK),
                "s_cselect_b32 s%d, 128, 0" % S_STEP,
                "s_cselect_b32 s%d, s%d, s%d" % (S_KL, S_TMP, S_KL),
                "s_cselect_b32 s%d, 1, 0" % S_TMP,
                "s_add_u32 s%d, s%d, s%d" % (S_WB, S_WB, S_STEP),
                "s_addc_u32 s%d, s%d, 0" % (S_WB + 1, S_WB + 1),
                "s_add_u32 s%d, s%d, s%d" % (S_SIT, S_SIT, S_TMP),
                "s_cmp_ge_u32 s%d, s%d" % (S_SIT, S_NKT),
                "s_cselect_b32 s%d, 0, s%d" % (S_SIT, S_SIT),
                "s_cselect_b32 s%d, 1024, 0" % S_STEP,
                "s_add_u32 s%d, s%d, s%d" % (S_TAPOFF, S_TAPOFF, S_STEP),
                "s_lshl_b32 s%d, s%d, 7" % (S_STEP, S_SIT),
                "s_add_u32 s%d, s%d, s%d" % (S_AB, S_XB2, S_STEP),
                "s_addc_u32 s%d, s%d, 0" % (S_AB + 1, S_XB2 + 1)]

    def plain(pieces):
        for m0w, d in pieces:
            e(m0w); e("s_nop 0"); e(d)

    # ---- setup
    e("s_mov_b64 s[%d:%d], %s" % (S_XB2, S_XB2 + 1, CWOP["xbase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_AB, S_AB + 1, CWOP["xbase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_WB, S_WB + 1, CWOP["wbase"]))
    e("s_mov_b32 s%d, %s" % (S_NK, CWOP["nk"]))
    e("s_mov_b32 s%d, %s" % (S_NKT, CWOP["nkt"]))
    e("s_mov_b32 s%d, %s" % (S_ADST, CWOP["adst"]))
    e("s_mov_b32 s%d, %s" % (S_WDST, CWOP["wdst"]))
    for sreg in (S_T, S_KL, S_SIT, S_TAPOFF):
        e("s_mov_b32 s%d, 0" % sreg)
    e("v_xor_b32_e32 %s, 64, %s" % (fa[("A", 1)], CWOP["faA0"]))
    e("v_xor_b32_e32 %s, 64, %s" % (fa[("W", 1)], CWOP["faW0"]))
    for r in range(c.NACC):
        e("v_accvgpr_write_b32 %s, 0" % ar(r))
    for t in table_reads():
        e(t)
    e("s_waitcnt lgkmcnt(0)")
    for t in table_adds():
        e(t)
    e("s_nop 1")
    plain(dma(0))
    for a in advance():
        e(a)
    for t in table_reads():
        e(t)
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")     # stage 0 landed AND its DMA has consumed the offset registers
    for t in table_adds():
        e(t)
    e("s_barrier")
    e("s_nop 1")
    plain(dma(1))
    first_entry_piece = (NM + spread - 1) // spread
    if first_entry_piece >= c.NA + c.NW:       # every piece is issued in the trailing block: entry0 re-issues none
        for a in advance():
            e(a)
    else:                                      # entry0 re-issues the pieces whose shadows lie behind it (same step, same offsets,
        e(dma(1)[first_entry_piece][0])        # same data: the activation offsets are still this step's), then advances
    for r in reads(0, 0, 0):
        e(r)
    e("s_branch %s" % ref("entry0"))

    def step(cur):
        pieces = dma(cur ^ 1)
        slots = [[] for _ in range(2 * NM)]
        dma_slots = [spread * j for j in range(len(pieces))]
        e(pieces[0][0])
        for j, (m0w, d) in enumerate(pieces):
            slots[dma_slots[j]].append(d)
            if j + 1 < len(pieces):
                slots[dma_slots[j]].append(pieces[j + 1][0])
        last = dma_slots[-1]
        for blk, (s32, fset) in enumerate(((0, 0), (1, 1))):
            rd = reads(cur, s32, fset)
            free = [i for i in range(blk * NM, (blk + 1) * NM) if i not in dma_slots]
            for i, r in enumerate(rd):
                slots[free[i]].append(r)
        tr = table_reads()                                     # block 1, behind its fragment reads and the loaders' advance
        used = [i for i in range(NM, 2 * NM) if slots[i]]
        base = max(used[-1], last) + 1
        assert base + 4 < 2 * NM
        slots[base] += tr[:3]
        for i in range(3, len(tr), 2):
            slots[base + 1 + (i - 3) // 2] += tr[i: i + 2]
        for blk in range(2):
            if blk == 1:
                lab("entry%d" % cur)
                e("s_waitcnt lgkmcnt(0)")
            mf = mfmas(1 if blk == 0 else 0)
            for i, m in enumerate(mf):
                e(m)
                for x in slots[blk * NM + i]:
                    e(x)
                if blk * NM + i == last:
                    for a in advance():
                        e(a)

    for k in range(2):
        lab("step%d" % k)
        step(k)
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        for t in table_adds():             # offsets of the step the loaders point at: this step's pieces were issued a block ago
            e(t)
        e("s_barrier")
        e("s_add_u32 s%d, s%d, 1" % (S_T, S_T))
        e("s_cmp_lt_u32 s%d, s%d" % (S_T, S_NK))
        e("s_cbranch_scc0 %s" % ref("exit"))
        if k == 1:
            e("s_branch %s" % ref("step0"))
    lab("exit")
    for m in mfmas(1):
        e(m)
    e("s_nop 15")
    e("s_nop 15")
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "open_sora_amd", "csrc"))
    ap.add_argument("--experiments", action="store_true",
                    help="also emit the bodies that lost their A/B runs in rounds 1-2 and are no longer shipped (one-tile-per-workgroup "
                         "bf16 GEMM, 256-wide / schedule-1 persistent 8-wave GEMM, 4-wave 32x32x16 GEMM and conv, 8-wave and per-tap "
                         "conv) -- into a scratch --out, never the shipped csrc")
    ap.add_argument("--ablations", action="store_true", help="with --experiments: the timing-only ablation bodies of the 4-wave GEMM")
    args = ap.parse_args()
    shipped_dir = os.path.realpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "open_sora_amd", "csrc"))
    if args.experiments and os.path.realpath(args.out) == shipped_dir:
        raise SystemExit("--experiments bodies are not shipped: give --out a scratch directory")

    def write_body(name, header, lines, shipped=True):
        if not shipped and not args.experiments:
            return
        with open(os.path.join(args.out, name), "w") as f:
            f.write("// GENERATED by tools/gen_gemm_asm.py -- do not edit.  %s\n" % header)
            for ln in lines():
                f.write('"%s\\n"\n' % ln)

    for bn in (256, 128):
        c = Cfg(bn)
        write_body("gemm256_body_n%d.inc" % bn, "256 x %d x 64 tile K loop." % bn, lambda: gen(c), shipped=False)
        with open(os.path.join(args.out, "gemm256_regs_n%d.inc" % bn), "w") as f:
            f.write("// GENERATED by tools/gen_gemm_asm.py -- do not edit.\n")
            P = "OSKG%d_" % bn
            f.write("#define %sSMEM %d\n#define %sW_BASE %d\n#define %sTM %d\n#define %sTN %d\n" % (P, c.SMEM, P, c.W_BASE, P, c.TM, P, c.TN))
            clob = ['"v%d"' % i for i in range(c.V0, c.V0 + c.VN)] + ['"a%d"' % i for i in range(c.NACC)] + \
                   ['"s%d"' % i for i in range(S_FIRST, S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']
            f.write("#define %sCLOBBERS %s\n" % (P, ", ".join(clob)))
            # accumulator tile t = AGPRs 16 t .. 16 t + 15 = quads 4 t .. 4 t + 3 (the wrappers bind them as asm outputs: acc_quads.h)
            f.write("#define %sACC_QUADS %d\n" % (P, c.NACC // 4))
            if args.experiments:
                sclob = ['"v%d"' % i for i in range(c.V0, c.V0 + c.VN + 4)] + ['"a%d"' % i for i in range(c.NACC)] + \
                        ['"s%d"' % i for i in range(S_FIRST, SEG_S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']
                f.write("#define %sSEG_CLOBBERS %s\n" % (P, ", ".join(sclob)))
                cclob = ['"v%d"' % i for i in range(c.V0, c.V0 + c.VN + 12)] + ['"a%d"' % i for i in range(c.NACC)] + \
                        ['"s%d"' % i for i in range(S_FIRST, CONV_S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']
                f.write("#define %sCONV_CLOBBERS %s\n" % (P, ", ".join(cclob)))
        for sched in (0, 1, 2) if bn == 256 else (0, 1):
            write_body("gemm256p_body_n%d_s%d.inc" % (bn, sched), "256 x %d x 64 tile K loop, persistent workgroup, schedule %d." % (bn, sched),
                       lambda: gen_pers(c, sched), shipped=(bn == 128 and sched == 0))
        if bn == 128 or args.experiments:
            with open(os.path.join(args.out, "gemm256p_regs_n%d.inc" % bn), "w") as f:
                f.write("// GENERATED by tools/gen_gemm_asm.py -- do not edit.\n")
                pclob = ['"v%d"' % i for i in range(c.V0, c.V0 + c.VN + 6)] + ['"a%d"' % i for i in range(c.NACC)] + \
                        ['"s%d"' % i for i in range(S_FIRST, PERS_S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']
                f.write("#define OSKP%d_CLOBBERS %s\n" % (bn, ", ".join(pclob)))
        write_body("conv256_body_n%d.inc" % bn, "256 x %d x 64 tile, whole K axis (all filter taps)." % bn, lambda: gen_conv(c), shipped=False)
        write_body("conv256_segment_n%d.inc" % bn, "256 x %d x 64 tile, one K segment (filter tap)." % bn, lambda: gen_segment(c), shipped=False)
    c = Cfg4()
    if args.ablations and args.experiments:   # timing-only experiments: never committed, never shipped
        for abl in (1, 2, 3, 4, 8, 12, 15, 16, 24, 31, 32, 56, 96):     # on the spread-2 schedule
            write_body("gemm256w_body_abl%d.inc" % abl, "timing experiment, WRONG RESULTS.", lambda: gen_w4(c, 0, abl, 0, 2), shipped=False)
    # the 4-wave 32x32x16 forms (LDS-DMA one piece per 2 shadows; other schedules: gen_w4(c, pf=1) L2 prefetch, dmak=1
    # buffer_load ... lds, spread=1 / 3, rd_per=1) were replaced by the 16x16x32 forms below
    write_body("gemm256w_body_spread2.inc", "256 x 256 x 64 tile, 4 waves x (128 x 128), persistent workgroup, one LDS-DMA piece per 2 MFMA shadows.",
               lambda: gen_w4(c, 0, 0, 0, 2), shipped=False)
    write_body("conv256w_body.inc", "256 x 256 x 64 tile, 4 waves x (128 x 128), whole K axis (all filter taps), one LDS-DMA piece per 2 MFMA shadows.",
               lambda: gen_conv_w4(c, 2), shipped=False)
    if args.experiments:
        with open(os.path.join(args.out, "gemm256w_regs.inc"), "w") as f:
            f.write("// GENERATED by tools/gen_gemm_asm.py -- do not edit.\n")
            f.write("#define OSKW_SMEM %d\n#define OSKW_W_BASE %d\n#define OSKW_TM %d\n#define OSKW_TN %d\n" % (c.SMEM, c.W_BASE, c.TM, c.TN))
            clob = ['"v%d"' % i for i in range(c.V0, c.V0 + c.VN + 8)] + ['"a%d"' % i for i in range(c.NACC)] + \
                   ['"s%d"' % i for i in range(S_FIRST, W4_S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']
            f.write("#define OSKW_CLOBBERS %s\n" % ", ".join(clob))
            cclob = ['"v%d"' % i for i in range(c.V0, c.V0 + c.VN + 6 + 1 + 2 * c.NA)] + ['"a%d"' % i for i in range(c.NACC)] + \
                    ['"s%d"' % i for i in range(S_FIRST, CONV_S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']
            f.write("#define OSKW_CONV_CLOBBERS %s\n" % ", ".join(cclob))
    cx = Cfg4x()
    write_body("gemm256x_body.inc", "256 x 256 x 64 tile, 4 waves x (128 x 128) on v_mfma_f32_16x16x32_bf16, persistent workgroup.", lambda: gen_x4(cx))
    write_body("conv256x_body.inc", "256 x 256 x 64 tile, 4 waves x (128 x 128) on v_mfma_f32_16x16x32_bf16, whole K axis (all filter taps).",
               lambda: gen_conv_x4(cx))
    cx128 = Cfg4x(4)
    write_body("conv256x_body_n128.inc", "256 x 128 x 64 tile, 4 waves x (128 x 64) on v_mfma_f32_16x16x32_bf16, whole K axis (all filter taps).",
               lambda: gen_conv_x4(cx128))
    with open(os.path.join(args.out, "gemm256x_regs.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_gemm_asm.py -- do not edit.\n")
        cclob = ['"v%d"' % i for i in range(cx128.V0, cx128.V0 + cx128.VN + 2 + 1 + 2 * cx128.NA)] + ['"a%d"' % i for i in range(cx128.NACC)] + \
                ['"s%d"' % i for i in range(S_FIRST, CONV_S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']
        f.write("#define OSKX128_SMEM %d\n#define OSKX128_CONV_CLOBBERS %s\n" % (cx128.SMEM, ", ".join(cclob)))
        f.write("#define OSKX_SMEM %d\n#define OSKX_W_BASE %d\n#define OSKX_NB %d\n" % (cx.SMEM, cx.W_BASE, cx.NB))
        clob = ['"v%d"' % i for i in range(cx.V0, cx.V0 + cx.VN + 2)] + ['"a%d"' % i for i in range(cx.NACC)] + \
               ['"s%d"' % i for i in range(S_FIRST, W4_S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']
        f.write("#define OSKX_CLOBBERS %s\n" % ", ".join(clob))
        cclob = ['"v%d"' % i for i in range(cx.V0, cx.V0 + cx.VN + 2 + 1 + 2 * cx.NA)] + ['"a%d"' % i for i in range(cx.NACC)] + \
                ['"s%d"' % i for i in range(S_FIRST, CONV_S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']
        f.write("#define OSKX_CONV_CLOBBERS %s\n" % ", ".join(cclob))
        # accumulator tile T = J * NB + I = AGPR quad T (the wrappers bind the quads as asm outputs: acc_quads.h)
        f.write("#define OSKX_ACC_QUADS %d\n#define OSKX128_ACC_QUADS %d\n" % (cx.NACC // 4, cx128.NACC // 4))
    for bn in (256, 128):
        c = Cfg(bn, fp8=True)
        with open(os.path.join(args.out, "gemm256_fp8_body_n%d.inc" % bn), "w") as f:
            f.write("// GENERATED by tools/gen_gemm_asm.py -- do not edit.  fp8 (e4m3) 256 x %d x 128 tile K loop.\n" % bn)
            for ln in gen_fp8(c):
                f.write('"%s\\n"\n' % ln)
        with open(os.path.join(args.out, "gemm256_fp8_regs_n%d.inc" % bn), "w") as f:
            f.write("// GENERATED by tools/gen_gemm_asm.py -- do not edit.\n")
            clob = ['"v%d"' % i for i in range(c.V0, c.V0 + c.VN)] + ['"a%d"' % i for i in range(c.NACC)] + \
                   ['"s%d"' % i for i in range(S_FIRST, S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']
            f.write("#define OSKQ%d_CLOBBERS %s\n" % (bn, ", ".join(clob)))


if __name__ == "__main__":
    main()
