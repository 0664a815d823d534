#!/usr/bin/env python
"""Generator of the hand-scheduled gfx950 main loop of the head_dim-72 flash-attention kernels
(open_sora_amd/csrc/attention_asm72.hip includes the emitted attention_asm72_n{NU}_v{VAR}.inc).

Why a generator: the MFMA shadow (32 cycles, ~5 issue slots) has to be filled by hand; hipcc's scheduler clusters
the softmax VALU work behind the MFMAs and shuffles accumulators between the VGPR and AGPR halves
(tools/isa_stream.py on attention_w64.hip shows it).  The schedule below is explicit and reproducible;
`python tools/gen_attn_asm.py --table NU` prints it shadow by shadow.

Two layouts of the same dataflow, workgroup = 256 query rows, KV tile = 64 keys:
  NU = 2: 4 waves x 64 rows (two 32-row query blocks u per wave share every K / V^T fragment), one wave per SIMD,
          392 registers per wave;
  NU = 1: 8 waves x 32 rows, two waves per SIMD (<= 256 registers each): twice the LDS fragment traffic, but each
          wave's fillers also sit in the shadow of its partner's MFMAs.
Dataflow (operand conventions of attention_w64.hip, the compiler-scheduled twin that validated the LDS images and
the pipeline on the GPU): S^T = K . Q^T and O^T += V^T . P^T on v_mfma_f32_32x32x16_bf16 with swapped operands (a
lane owns one query).  Q is pre-multiplied by scale*log2(e); the reference max M (bf16-exact, log2 units) sits,
negated, in Q's padding dim 72 and K's padding dim 72 reads 1.0 from a constant LDS chunk, so the MFMA delivers
S' = q.k - M and P = exp2(S') needs ONE v_exp per score; row sums come out of the P.V MFMA (ones row of V^T,
accumulator row 72).  M only moves when some score exceeds it by more than 2^THR (rare path: rescale O, shift the
pending scores, rewrite the padding dim).
Body t (starts right after barrier t-1): the last 2 fragment pairs' P.V MFMAs of tile t-1 | QK^T of tile t+1 |
P.V of tile t (first 10 of 12 fragment pairs); beside them: K / V^T fragment reads (4-deep rings), LDS-DMA of
K(t+2), V(t+1), exp2 + pack of tile t, max of tile t+1.
"""
import argparse
import os

HD, NKS, NDT = 72, 5, 3
KTILE, VTILE = 9216, 12288
KOFF = [0, KTILE]
VOFF = [2 * KTILE, 2 * KTILE + VTILE]
CONST_OFF = 2 * KTILE + 2 * VTILE          # 16-byte chunk {1.0bf16, 0...}: K's padding dims 72..79, one copy per K ring
SMEM = CONST_OFF + KTILE + 16              # slot at the same distance (KTILE) as the slots: one address register serves both
THR_BITS = "0x41000000"                    # 8.0: move M when a score exceeds the reference by > 2^8
NKD = NVD = 9                              # LDS-DMA wave instructions per K / V^T tile

# ---- asm-owned SGPRs
S_FIRST, S_LAST = 36, 65
S_KRG, S_VRG, S_FLG, S_NRG = 36, 38, 45, 64  # lane masks: K / V^T loader sits on a ragged (segment-last) tile; flags
# (S_FLG bit 1: this wave maintains the ones rows; S_NRG = 1 when the launch has NO ragged tile)
S_KB, S_VB, S_KSTEP, S_KJ, S_VJ = 40, 42, 44, 46, 48
S_TPS, S_NT, S_KDST, S_VDST, S_NKW, S_NVW = 50, 51, 52, 53, 54, 55
S_T, S_KTT, S_VTT, S_TMP, S_HIM, S_KL, S_VL = 56, 57, 58, 59, 60, 62, 63

# ---- asm operands (order = operand numbers in the wrapper's asm statement)
OPERANDS = ["m0out", "m1out",
            "koff0", "koff1", "koff2", "voff0", "voff1", "voff2",
            "fo0", "fo1", "fo2", "fo3", "kc0", "kc1", "koffL0", "koffL1", "koffL2", "maskval", "onesaddr",
            "kbase", "vbase", "kstep", "kjump", "vjump", "tps", "nt", "kdst", "vdst", "nkw", "nvw"]
OP = {n: "%%%d" % i for i, n in enumerate(OPERANDS)}


def vr(base, n=1):
    return "v%d" % base if n == 1 else "v[%d:%d]" % (base, base + n - 1)


def ar(base, n=1):
    return "a%d" % base if n == 1 else "a[%d:%d]" % (base, base + n - 1)


class Layout:
    """register file and schedule geometry for NU query blocks per wave"""

    def __init__(self, nu):
        self.NU = nu
        self.NW = 8 // nu                      # waves per workgroup
        self.NSLOT = (NKD + self.NW - 1) // self.NW   # LDS-DMA slots per wave and tile (last one conditional)
        self.V_FIRST = 48
        self.SA0 = 48
        self.SB0 = self.SA0 + 32 * nu
        self.PB0 = self.SB0 + 32 * nu
        self.KR0 = self.PB0 + 16 * nu          # fragment rings, 4 slots x 4 registers each
        self.VR0 = self.KR0 + 16
        self.TMP0 = self.VR0 + 16              # 8 temporaries
        self.MT = [self.TMP0 + 8, self.TMP0 + 9]
        self.MM = [self.TMP0 + 10, self.TMP0 + 11]
        self.TX = [self.TMP0 + 12 + i for i in range(4)]
        self.V_END = self.TMP0 + 16
        self.A_O0 = 0
        self.A_Q0 = 16 * NDT * nu
        self.A_END = self.A_Q0 + 4 * NKS * nu
        self.NTRAIL = 2 * nu                   # MFMAs of the 2 trailing fragment pairs
        self.NQK = 10 * nu
        self.NPVB = 10 * nu                    # P.V MFMAs inside the body
        self.I_QK0 = self.NTRAIL               # global shadow index of the first QK^T MFMA
        # first / last shadow of filler classes A (exp2+pack of key groups 0..2), B (group 3), C (max of tile t+1)
        self.WINDOWS = [4, 25, 24, 40, 26, 43] if nu == 2 else [2, 13, 12, 19, 14, 21]
        self.C_T2_RELEASE = 28 if nu == 2 else 15   # chains over keys 32..63: their last QK^T MFMAs come last

    def S(self, setbase, u, t2, r=0):
        return setbase + (u * 2 + t2) * 16 + r

    def PB(self, u, g, w=0):
        return self.PB0 + (u * 4 + g) * 4 + w

    def AO(self, u, d):
        return self.A_O0 + (u * NDT + d) * 16

    def AQ(self, u, ks):
        return self.A_Q0 + (u * NKS + ks) * 4


class Stream:
    """instruction list with in-order LDS-read bookkeeping (lgkmcnt)"""

    def __init__(self, ablate=frozenset()):
        self.lines, self.pending, self.table = [], [], []
        self.ablate, self.in_body = ablate, False

    def emit(self, text, kind="x"):
        ab = self.ablate if self.in_body else frozenset()
        if ("noexp" in ab and text.startswith("v_exp")) or ("nobar" in ab and text.startswith("s_barrier")) or \
           ("nodma" in ab and text.startswith("global_load_lds")) or \
           ("nolds" in ab and (text.startswith("ds_read") or "lgkmcnt" in text)) or \
           ("novalu" in ab and kind in ("e", "v") and not text.startswith("v_mfma") and not text.startswith("v_cmp")) or \
           ("nomfma" in ab and text.startswith("v_mfma")) or ("norare" in ab and text.startswith("s_cbranch_vccnz")) or \
           ("nocvt" in ab and text.startswith("v_cvt_pk")) or ("nomax" in ab and (text.startswith("v_max") or text.startswith("v_cmp"))):
            return
        self.lines.append("  " + text)
        self.table.append((kind, text))

    def label(self, name):
        self.lines.append(name + ":")
        self.table.append(("L", name))

    def ds_read(self, dst, addr_op, imm, tag):
        self.emit("ds_read_b128 %s, %s offset:%d" % (vr(dst, 4), addr_op, imm), "d")
        self.pending.append(tag)

    def need(self, tag):
        """wait until the read `tag` has landed (reads return in order)"""
        if tag in self.pending:
            idx = self.pending.index(tag)
            self.emit("s_waitcnt lgkmcnt(%d)" % (len(self.pending) - 1 - idx), "w")
            self.pending = self.pending[idx + 1:]


# ------------------------------------------------------------------------------------------ pieces
def k_read(st, L, slot, p, tag):
    """K fragment of pair p = (ks, t2) of the tile in ring slot `slot` -> K ring"""
    ks, t2 = p // 2, p % 2
    dst = L.KR0 + (p % 4) * 4
    if ks < 4:
        st.ds_read(dst, OP["fo%d" % ks], KOFF[slot] + t2 * 4096, tag)
    else:
        st.ds_read(dst, OP["kc%d" % t2], KOFF[slot], tag)


def v_read(st, L, slot, r, tag):
    g, d = r // NDT, r % NDT
    st.ds_read(L.VR0 + (r % 4) * 4, OP["fo%d" % g], VOFF[slot] + d * 4096, tag)


def qk_mfma(st, L, sn, a):
    p, u = a // L.NU, a % L.NU
    ks, t2 = p // 2, p % 2
    dst = vr(L.S(sn, u, t2), 16)
    st.emit("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (dst, vr(L.KR0 + (p % 4) * 4, 4), ar(L.AQ(u, ks), 4),
                                                       "0" if ks == 0 else dst), "M")


def pv_mfma(st, L, b):
    r, u = b // L.NU, b % L.NU
    g, d = r // NDT, r % NDT
    dst = ar(L.AO(u, d), 16)
    st.emit("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (dst, vr(L.VR0 + (r % 4) * 4, 4), vr(L.PB(u, g), 4), dst), "M")


def exp_group(L, sc, u, g):
    """exp2 + pack of 8 scores (16 keys of one query block) -> one B-operand fragment of P"""
    t2, r0 = g >> 1, (g & 1) * 8
    ops = []
    for r in range(8):
        x = vr(L.S(sc, u, t2, r0 + r))
        ops.append(("e", "v_exp_f32 %s, %s" % (x, x)))
    for w in range(4):
        ops.append(("v", "v_cvt_pk_bf16_f32 %s, %s, %s" % (vr(L.PB(u, g, w)), vr(L.S(sc, u, t2, r0 + 2 * w)),
                                                           vr(L.S(sc, u, t2, r0 + 2 * w + 1)))))
    return ops


def max_chain(L, sn, u, t2, tmp):
    x = lambda r: vr(L.S(sn, u, t2, r))
    ops = [("v", "v_max3_f32 %s, %s, %s, %s" % (vr(tmp), x(0), x(1), x(2)))]
    for k in range(6):
        ops.append(("v", "v_max3_f32 %s, %s, %s, %s" % (vr(tmp), vr(tmp), x(3 + 2 * k), x(4 + 2 * k))))
    ops.append(("v", "v_max_f32 %s, %s, %s" % (vr(tmp), vr(tmp), x(15))))
    return ops


def chain_tmp(L, u, t2):
    return L.TMP0 + t2 * L.NU + u


def lanemax_final(L):
    """common path: ONE number per lane = max of its pending scores; the exact per-row max (other half-wave included)
    is only formed in the rare path.  The compare sits here so that VCC is old news at the next body's branch."""
    t = [vr(L.TMP0 + i) for i in range(2 * L.NU)]
    if L.NU == 2:
        ops = [("v", "v_max3_f32 %s, %s, %s, %s" % (vr(L.MT[0]), t[0], t[1], t[2])),
               ("v", "v_max_f32 %s, %s, %s" % (vr(L.MT[0]), vr(L.MT[0]), t[3]))]
    else:
        ops = [("v", "v_max_f32 %s, %s, %s" % (vr(L.MT[0]), t[0], t[1]))]
    return ops + [("v", "v_cmp_lt_f32 vcc, %s, %s" % (THR_BITS, vr(L.MT[0])))]


def rowmax_from_chains(st, L):
    """rare path / prologue: per query block, combine the two chain maxima and the other half-wave's -> MT[u]"""
    for u in range(L.NU):
        ta, tb = chain_tmp(L, u, 0), chain_tmp(L, u, 1)
        st.emit("v_max_f32 %s, %s, %s" % (vr(ta), vr(ta), vr(tb)))
        st.emit("v_mov_b32 %s, %s" % (vr(tb), vr(ta)))
        st.emit("s_nop 1", "n")
        st.emit("v_permlane32_swap_b32 %s, %s" % (vr(ta), vr(tb)))
        st.emit("s_nop 1", "n")
        st.emit("v_max_f32 %s, %s, %s" % (vr(L.MT[u]), vr(ta), vr(tb)))


def k_dma(st, L, slot, i, part=3):
    """K loader slot i of this wave -> ring slot `slot` (instruction j = wave + NW i).  part 1 = M0 write only,
    2 = the DMA only (one other instruction must sit between them), 3 = both with an s_nop.  On the ragged last tile
    of a key segment (lane mask S_KRG) the rows past the segment re-fetch its last key (offsets koffL)."""
    if part & 1:
        st.emit("s_add_u32 m0, s%d, %d" % (S_KDST, KOFF[slot] + 1024 * L.NW * i), "s")
        st.emit("v_cndmask_b32_e64 %s, %s, %s, s[%d:%d]" % (vr(L.TX[3]), OP["koff%d" % i], OP["koffL%d" % i], S_KRG, S_KRG + 1), "v")
    if part & 2:
        st.emit("global_load_lds_dwordx4 %s, s[%d:%d]" % (vr(L.TX[3]), S_KB, S_KB + 1), "g")


def v_dma(st, L, slot, i, part=3):
    if part & 1:
        st.emit("s_add_u32 m0, s%d, %d" % (S_VDST, (VOFF[slot] - VOFF[0]) + 1024 * L.NW * i), "s")
    if part == 3:
        st.emit("s_nop 0", "n")
    if part & 2:
        st.emit("global_load_lds_dwordx4 %s, s[%d:%d]" % (OP["voff%d" % i], S_VB, S_VB + 1), "g")


def dma_last(st, L, which, slot, uid):
    """last loader slot: only the wave(s) whose instruction index is < 9"""
    lab = ".L@@_%s2%s" % (which, uid)
    st.emit("s_cmp_lt_u32 s%d, %d" % (S_NKW if which == "k" else S_NVW, L.NSLOT), "s")
    st.emit("s_cbranch_scc1 %s" % lab, "s")
    (k_dma if which == "k" else v_dma)(st, L, slot, L.NSLOT - 1)
    st.label(lab)


def advance(st, which, uid):
    """point the loader at its next tile; past the last tile it stays (harmless re-fetch of the last tile)"""
    lab = ".L@@_%sa%s" % (which, uid)
    sl, sb, stt, sj = (S_KL, S_KB, S_KTT, S_KJ) if which == "k" else (S_VL, S_VB, S_VTT, S_VJ)
    st.emit("s_add_u32 s%d, s%d, 1" % (S_TMP, sl), "s")
    st.emit("s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NT), "s")
    st.emit("s_cbranch_scc0 %s" % lab, "s")
    st.emit("s_mov_b32 s%d, s%d" % (sl, S_TMP), "s")
    if which == "k":
        st.emit("s_add_u32 s%d, s%d, s%d" % (sb, sb, S_KSTEP), "s")
    else:
        st.emit("s_add_u32 s%d, s%d, 128" % (sb, sb), "s")
    st.emit("s_addc_u32 s%d, s%d, 0" % (sb + 1, sb + 1), "s")
    st.emit("s_add_u32 s%d, s%d, 1" % (stt, stt), "s")
    st.emit("s_cmp_lg_u32 s%d, s%d" % (stt, S_TPS), "s")
    st.emit("s_cbranch_scc1 %s" % lab, "s")
    st.emit("s_mov_b32 s%d, 0" % stt, "s")
    st.emit("s_add_u32 s%d, s%d, s%d" % (sb, sb, sj), "s")
    st.emit("s_addc_u32 s%d, s%d, s%d" % (sb + 1, sb + 1, sj + 1), "s")
    st.label(lab)


def ragged_masks(st):
    """refresh both loaders' ragged-tile lane masks; skipped entirely when the launch has no ragged tile"""
    lab = ".L@@_rg%d" % len(st.lines)
    st.emit("s_cmp_lg_u32 s%d, 0" % S_NRG, "s")
    st.emit("s_cbranch_scc1 %s" % lab, "s")
    ragged_mask(st, "k")
    ragged_mask(st, "v")
    st.label(lab)


def ragged_mask(st, which):
    """lane mask (all ones / zero): the loader's current tile is the last one of its key segment AND that tile is
    ragged (flag bit 0)"""
    stt, srg = (S_KTT, S_KRG) if which == "k" else (S_VTT, S_VRG)
    st.emit("s_add_u32 s%d, s%d, 1" % (S_TMP, stt), "s")
    st.emit("s_sub_u32 s%d, s%d, s%d" % (S_TMP, S_TPS, S_TMP), "s")        # 0 on the segment's last tile
    st.emit("s_or_b32 s%d, s%d, s%d" % (S_TMP, S_TMP, S_NRG), "s")         # never 0 when the launch has no ragged tile
    st.emit("s_cmp_eq_u32 s%d, 0" % S_TMP, "s")
    st.emit("s_cselect_b64 s[%d:%d], -1, 0" % (srg, srg + 1), "s")


def ones_row(st, L, slot):
    """V^T ring slot `slot` is about to receive the tile the V^T loader points at: its ones row (accumulator row 72 =
    softmax denominator) becomes that tile's key-validity mask (all ones unless the tile is ragged).  Wave 0 only
    (flag bit 1); lanes 32..63 rewrite the zero row behind it."""
    lab = ".L@@_or%d_%d" % (slot, len(st.lines))
    st.emit("s_bitcmp1_b32 s%d, 1" % S_FLG, "s")
    st.emit("s_cbranch_scc0 %s" % lab, "s")
    st.emit("v_mov_b32 %s, 0x3f803f80" % vr(L.TX[2]), "v")
    st.emit("v_cndmask_b32_e64 %s, %s, 0, s[%d:%d]" % (vr(L.TX[2]), vr(L.TX[2]), S_HIM, S_HIM + 1), "v")
    st.emit("v_cndmask_b32_e64 %s, %s, %s, s[%d:%d]" % (vr(L.TX[2]), vr(L.TX[2]), OP["maskval"], S_VRG, S_VRG + 1), "v")
    st.emit("ds_write_b32 %s, %s offset:%d" % (OP["onesaddr"], vr(L.TX[2]), VOFF[slot] - VOFF[0]), "D")
    st.label(lab)


def dma_group(st, L, which, slot, uid):
    """all LDS-DMA instructions of this wave for one K (or V^T) tile + loader advance (prologue form)"""
    for i in range(L.NSLOT - 1):
        (k_dma if which == "k" else v_dma)(st, L, slot, i)
    dma_last(st, L, which, slot, uid)
    advance(st, which, uid)
    ragged_masks(st)


def fixup(st, L, sx, init):
    """move the reference max M to (M + max(mt, 0)) [init: to mt], rounded to bf16: shift the pending scores,
    rescale O (not at init: O == 0), rewrite Q's padding dim 72 with -M."""
    if not init:
        st.emit("s_nop 15", "n")
        st.emit("s_nop 15", "n")  # trailing P.V MFMAs -> v_accvgpr_read of O
    rowmax_from_chains(st, L)
    T = L.TMP0
    for u in range(L.NU):
        d, n, pk, f, de, al, t = L.TX[0], L.TX[1], L.TX[2], L.TX[3], T + 4, T + 5, T + 6
        if init:
            st.emit("v_mov_b32 %s, %s" % (vr(n), vr(L.MT[u])))
        else:
            st.emit("v_max_f32 %s, 0, %s" % (vr(d), vr(L.MT[u])))
            st.emit("v_add_f32 %s, %s, %s" % (vr(n), vr(L.MM[u]), vr(d)))
        st.emit("v_xor_b32 %s, 0x80000000, %s" % (vr(n), vr(n)))              # -(M + mt')
        st.emit("v_cvt_pk_bf16_f32 %s, %s, 0" % (vr(pk), vr(n)))                # lo16 = bf16(-M_new), hi16 = 0
        st.emit("v_lshlrev_b32 %s, 16, %s" % (vr(f), vr(pk)))                   # f32(-M_new)
        st.emit("v_add_f32 %s, %s, %s" % (vr(de), vr(L.MM[u]), vr(f)))          # M_old - M_new (<= 0 in the loop)
        st.emit("v_xor_b32 %s, 0x80000000, %s" % (vr(L.MM[u]), vr(f)))          # M = M_new
        if not init:
            st.emit("v_exp_f32 %s, %s" % (vr(al), vr(de)))                      # alpha = 2^(M_old - M_new)
        # Q padding dim 72 lives in lanes 32..63 of word 0 of the k-step-4 fragment
        st.emit("v_accvgpr_read_b32 %s, %s" % (vr(t), ar(L.AQ(u, 4))))
        st.emit("s_nop 0", "n")
        st.emit("v_cndmask_b32_e64 %s, %s, %s, s[%d:%d]" % (vr(t), vr(t), vr(pk), S_HIM, S_HIM + 1))
        st.emit("s_nop 0", "n")
        st.emit("v_accvgpr_write_b32 %s, %s" % (ar(L.AQ(u, 4)), vr(t)))
        for t2 in range(2):
            for r in range(16):
                x = vr(L.S(sx, u, t2, r))
                st.emit("v_add_f32 %s, %s, %s" % (x, x, vr(de)))
        if not init:
            for dd in range(NDT):
                for r0 in range(0, 16, 4):
                    for r in range(r0, r0 + 4):
                        st.emit("v_accvgpr_read_b32 %s, %s" % (vr(T + (r - r0)), ar(L.AO(u, dd) + r)))
                    st.emit("s_nop 0", "n")
                    for r in range(r0, r0 + 4):
                        st.emit("v_mul_f32 %s, %s, %s" % (vr(T + (r - r0)), vr(T + (r - r0)), vr(al)))
                    st.emit("s_nop 0", "n")
                    for r in range(r0, r0 + 4):
                        st.emit("v_accvgpr_write_b32 %s, %s" % (ar(L.AO(u, dd) + r), vr(T + (r - r0))))
    st.emit("s_nop 7", "n")  # v_accvgpr_write -> MFMA operand


# ------------------------------------------------------------------------------------------ body
def body(st, L, k, safe):
    """iteration with ring slot parity k: SC = scores of tile t (k == 0: set A), SN receives tile t+1"""
    NU = L.NU
    sc, sn = (L.SA0, L.SB0) if k == 0 else (L.SB0, L.SA0)
    cur = k
    uid = "b%d" % k
    st.label(".L@@_body%d" % k)
    # -- top: K fragment reads of pairs 0..3 (tile t+1 sits in ring slot cur^1)
    for p in range(4):
        k_read(st, L, cur ^ 1, p, ("k", p))
    # -- trailing P.V MFMAs of tile t-1 (fragment pairs 10, 11 were read before the barrier); in their shadows the
    #    first LDS-DMA pieces of K(t+2) -> slot cur and V(t+1) -> slot cur^1 (M0 write BEFORE the MFMA: no s_nop)
    pieces = [("k", i) for i in range(L.NSLOT - 1)] + [("v", i) for i in range(L.NSLOT - 1)]
    pieces = pieces[:L.NTRAIL]
    done = {"k": sum(1 for w, _ in pieces if w == "k"), "v": sum(1 for w, _ in pieces if w == "v")}

    def piece(pc, part):
        (k_dma if pc[0] == "k" else v_dma)(st, L, cur if pc[0] == "k" else cur ^ 1, pc[1], part)

    for n in range(L.NTRAIL):
        if n < len(pieces):
            piece(pieces[n], 1)
        pv_mfma(st, L, 10 * NU + n)
        if n < len(pieces):
            piece(pieces[n], 2)
    # -- decision: did some score of tile t exceed the reference by more than 2^THR (VCC from the previous body)?
    st.emit("s_cbranch_vccnz .L@@_rare%d" % k)
    st.label(".L@@_entry%d" % k)

    # -- remaining LDS-DMA work, one item per shadow right after the entry point: (emitter, cycles)
    later = []
    for i in range(done["k"], L.NSLOT - 1):
        later.append((lambda i=i: k_dma(st, L, cur, i), 12))
    later.append((lambda: dma_last(st, L, "k", cur, uid), 12))
    for i in range(done["v"], L.NSLOT - 1):
        later.append((lambda i=i: v_dma(st, L, cur ^ 1, i), 12))
    later.append((lambda: dma_last(st, L, "v", cur ^ 1, uid), 12))
    later.append((lambda: ones_row(st, L, cur ^ 1), 8))       # before the V^T loader moves on: S_VRG is tile t+1's
    later.append((lambda: advance(st, "k", uid), 28))
    later.append((lambda: (advance(st, "v", uid), ragged_masks(st)), 32))

    # -- fillers paced in CYCLES (v_exp 8, other VALU 4): three classes, each spread uniformly over its window
    cost = lambda kind_: 8.0 if kind_ == "e" else (2.0 if kind_ == "n" else 4.0)
    clsA, clsB, clsC = [], [], []
    for g in range(4):
        for u in range(NU):
            (clsA if g < 3 else clsB).extend(exp_group(L, sc, u, g))
    ca = [max_chain(L, sn, u, 0, chain_tmp(L, u, 0)) for u in range(NU)]
    cb = [max_chain(L, sn, u, 1, chain_tmp(L, u, 1)) for u in range(NU)]
    clsC.extend([o for grp in zip(*ca) for o in grp])
    n_first_b = len(clsC)
    clsC.extend([o for grp in zip(*cb) for o in grp])
    clsC.extend(lanemax_final(L))
    W = L.WINDOWS
    classes = [[clsA, W[0], W[1], 0, 0.0], [clsB, W[2], W[3], 0, 0.0], [clsC, W[4], W[5], 0, 0.0]]
    totals = [sum(cost(o[0]) for o in c[0]) for c in classes]
    mf = [("qk", a) for a in range(L.NQK)] + [("pv", b) for b in range(L.NPVB)]
    for n, (kind, idx) in enumerate(mf):
        i = L.I_QK0 + n
        pair, u = idx // NU, idx % NU
        if u == 0 and pair % 2 == 0:     # pairs 2m and 2m+1 with one wait (both were issued >= 3 pairs ago)
            st.need(("k" if kind == "qk" else "v", pair + 1))
        if kind == "qk":
            qk_mfma(st, L, sn, idx)
        else:
            pv_mfma(st, L, idx)
        if safe:
            st.emit("s_nop 7", "n")
        used = 0.0
        # ring reads: after the last MFMA of pair x its ring slot is free -> read pair x + 4
        if u == NU - 1:
            if kind == "qk":
                if pair + 4 < 10:
                    k_read(st, L, cur ^ 1, pair + 4, ("k", pair + 4))
                else:
                    v_read(st, L, cur, pair + 4 - 10, ("v", pair + 4 - 10))   # V pairs 0..3 behind the last K pairs
                used += 4
            elif pair + 4 < 12:
                v_read(st, L, cur, pair + 4, ("v", pair + 4))
                used += 4
        if later:
            fn, cyc = later.pop(0)
            fn()
            used += cyc
        for ci, c in enumerate(classes):
            items, first, last = c[0], c[1], c[2]
            if i < first:
                continue
            frac = min(1.0, (i - first + 1) / float(last - first + 1))
            while c[3] < len(items):
                kind_, text = items[c[3]]
                if ci == 2 and c[3] >= n_first_b and i < L.C_T2_RELEASE:
                    break
                if totals[ci] * frac - c[4] <= 0 and i < last:
                    break
                if used >= 30 and i < last:
                    break
                st.emit(text, kind_)
                used += cost(kind_)
                c[4] += cost(kind_)
                c[3] += 1
    for c in classes:
        assert c[3] == len(c[0]), "unscheduled filler work"
    assert not later
    # -- end of body: DMA landed + every fragment read retired, then the tile barrier
    st.emit("s_waitcnt vmcnt(0) lgkmcnt(0)", "w")
    st.pending = []
    st.emit("s_barrier", "B")
    st.emit("s_add_u32 s%d, s%d, 1" % (S_T, S_T), "s")
    st.emit("s_cmp_lt_u32 s%d, s%d" % (S_T, S_NT), "s")
    st.emit("s_cbranch_scc0 .L@@_exit", "s")
    if k == 1:
        st.emit("s_branch .L@@_body0", "s")
    return pieces


def generate(L, safe=False, ablate=frozenset()):
    st = Stream(ablate)
    e = st.emit
    # ---- copy the mutable scalars into asm-owned SGPRs
    e("s_mov_b64 s[%d:%d], %s" % (S_KB, S_KB + 1, OP["kbase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_VB, S_VB + 1, OP["vbase"]))
    e("s_mov_b32 s%d, %s" % (S_KSTEP, OP["kstep"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_KJ, S_KJ + 1, OP["kjump"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_VJ, S_VJ + 1, OP["vjump"]))
    for sreg, name in ((S_TPS, "tps"), (S_NT, "nt"), (S_KDST, "kdst"), (S_VDST, "vdst"), (S_NKW, "nkw"), (S_NVW, "nvw")):
        e("s_mov_b32 s%d, %s" % (sreg, OP[name]))
    for sreg in (S_T, S_KTT, S_VTT, S_KL, S_VL, S_HIM):
        e("s_mov_b32 s%d, 0" % sreg)
    e("s_mov_b32 s%d, -1" % (S_HIM + 1))
    # nvw = valid V^T loader slots | (no ragged tile in this launch) << 8 | (this wave maintains the ones rows) << 9
    e("s_lshr_b32 s%d, s%d, 8" % (S_FLG, S_NVW))
    e("s_and_b32 s%d, s%d, 1" % (S_NRG, S_FLG))
    e("s_and_b32 s%d, s%d, 0xff" % (S_NVW, S_NVW))
    ragged_mask(st, "k")
    ragged_mask(st, "v")
    for u in range(2):
        e("v_mov_b32 %s, 0" % vr(L.MM[u]))
    for r in range(L.A_O0, L.A_Q0):
        e("v_accvgpr_write_b32 %s, 0" % ar(r))
    # ---- prologue: K0, V0 -> slot 0, K1 -> slot 1
    dma_group(st, L, "k", 0, "p0")
    dma_group(st, L, "v", 0, "p1")
    dma_group(st, L, "k", 1, "p2")
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    # scores of tile 0 -> set A (Q's padding dim is 0: raw scores)
    for p in range(10):
        k_read(st, L, 0, p, ("k", p))
        st.need(("k", p))
        for u in range(L.NU):
            qk_mfma(st, L, L.SA0, p * L.NU + u)
    e("s_barrier")                      # every wave has read K0: slot 0 may be refilled
    e("s_nop 15")
    e("s_nop 15")
    for t2 in range(2):
        for u in range(L.NU):
            for op in max_chain(L, L.SA0, u, t2, chain_tmp(L, u, t2)):
                e(op[1])
    fixup(st, L, L.SA0, init=True)
    # what body 0 does before its entry point: the first LDS-DMA pieces of K(2) -> slot 0 and V(1) -> slot 1 (the
    # same ones body() puts into the trailing shadows), the first K fragment reads of tile 1
    pieces = body(Stream(), L, 0, False)
    for w, i in pieces:
        (k_dma if w == "k" else v_dma)(st, L, 0 if w == "k" else 1, i)
    for p in range(4):
        k_read(st, L, 1, p, ("k", p))
    e("s_branch .L@@_entry0")
    # ---- the two loop bodies
    st.in_body = True
    for k in range(2):
        st.pending = []
        body(st, L, k, safe)
    st.in_body = False
    # ---- rare paths
    for k in range(2):
        st.label(".L@@_rare%d" % k)
        fixup(st, L, L.SA0 if k == 0 else L.SB0, init=False)
        e("s_branch .L@@_entry%d" % k)
    # ---- exit: the trailing P.V MFMAs of the last tile
    st.label(".L@@_exit")
    for n in range(L.NTRAIL):
        pv_mfma(st, L, 10 * L.NU + n)
    e("s_nop 15")
    e("s_nop 15")
    e("v_mov_b32 %s, %s" % (OP["m0out"], vr(L.MM[0])))
    e("v_mov_b32 %s, %s" % (OP["m1out"], vr(L.MM[1])))
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table", type=int, default=0, help="print the production schedule of layout NU (1 or 2)")
    ap.add_argument("--exp", default="safe", help="experimental variant 1: safe | ablations joined by + (noexp nobar "
                    "nodma nolds novalu nomfma norare nocvt nomax): timing only, wrong results")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "open_sora_amd", "csrc"))
    args = ap.parse_args()
    for nu in (2, 1):
        L = Layout(nu)
        exp_safe = args.exp == "safe"
        exp_ab = frozenset() if exp_safe else frozenset(args.exp.split("+"))
        for vi, (safe, ablate) in enumerate([(False, frozenset()), (exp_safe, exp_ab)]):
            st = generate(L, safe, ablate)
            if args.table == nu and vi == 0:
                gap = []
                for kind, text in st.table:
                    if kind == "M":
                        print("".join(gap)); gap = ["M "]
                    elif kind == "L":
                        print("".join(gap)); gap = []; print(text + ":")
                    else:
                        gap.append({"x": "v"}.get(kind, kind))
                print("".join(gap))
            with open(os.path.join(args.out, "attention_asm72_n%d_v%d.inc" % (nu, vi)), "w") as f:
                f.write("// GENERATED by tools/gen_attn_asm.py -- do not edit.  layout NU=%d, variant %d: %s\n" %
                        (nu, vi, "production" if vi == 0 else ("hazard-padded (debug)" if safe else "timing ablation " + "+".join(sorted(ablate)))))
                for ln in st.lines:
                    f.write('"%s\\n"\n' % ln.replace("@@", "osk72n%dv%d" % (nu, vi)))
    # register / operand contract for the wrapper
    with open(os.path.join(args.out, "attention_asm72_regs.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_attn_asm.py -- do not edit.\n")
        f.write("#define OSK72_SMEM %d\n#define OSK72_CONST_OFF %d\n" % (SMEM, CONST_OFF))
        f.write("#define OSK72_KTILE %d\n#define OSK72_VTILE %d\n#define OSK72_VOFF0 %d\n" % (KTILE, VTILE, VOFF[0]))
        for nu in (2, 1):
            L = Layout(nu)
            P = "OSK72N%d_" % nu
            clob = ['"v%d"' % i for i in range(L.V_FIRST, L.V_END)] + ['"a%d"' % i for i in range(0, L.A_END)] + \
                   ['"s%d"' % i for i in range(S_FIRST, S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']
            f.write("#define %sCLOBBERS %s\n" % (P, ", ".join(clob)))
            f.write("#define %sA_CLOBBERS %s\n" % (P, ", ".join('"a%d"' % i for i in range(0, L.A_END))))
            f.write("#define %sNSLOT %d\n" % (P, L.NSLOT))
            for u in range(nu):   # Q fragment words of query block u (operands %0..%19) -> AGPRs
                f.write("#define %sQW%d %s\n" % (P, u, " ".join('"v_accvgpr_write_b32 a%d, %%%d\\n"' % (L.AQ(u, 0) + i, i) for i in range(NKS * 4))))
            for u in range(nu):   # O^T row tile (u, d) -> operands %0..%15
                for d in range(NDT):
                    f.write("#define %sOR%d %s\n" % (P, u * NDT + d, " ".join('"v_accvgpr_read_b32 %%%d, a%d\\n"' % (i, L.AO(u, d) + i) for i in range(16))))


if __name__ == "__main__":
    main()
