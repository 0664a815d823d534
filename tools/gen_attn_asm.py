#!/usr/bin/env python
"""Generator of the hand-scheduled gfx950 main loop of the head_dim-72 flash-attention kernel
(open_sora_amd/csrc/attention_asm72.hip includes the emitted attention_asm72_body.inc).

Why a generator: one wave per SIMD owns the whole 512-register file, so every MFMA shadow (32 cycles, ~5 issue
slots) has to be filled by hand; hipcc's scheduler clusters the softmax VALU work behind the MFMAs and shuffles
accumulators between the VGPR and AGPR halves (tools/isa_stream.py on attention_w64.hip shows it).  The
schedule below is explicit and reproducible; `python tools/gen_attn_asm.py --table` prints it gap by gap.

Dataflow (same operand conventions as attention_w64.hip, which is the compiler-scheduled twin used to validate
the LDS images and the pipeline on the GPU):
  wave = 64 query rows = 2 query blocks u; KV tile = 64 keys; S^T = K . Q^T and O^T += V^T . P^T on
  v_mfma_f32_32x32x16_bf16 with swapped operands (a lane owns one query).
  Q is pre-multiplied by scale*log2(e); the running max M (bf16-exact, log2 units) sits, negated, in Q's padding
  dim 72 and K's padding dim 72 reads 1.0 from a constant LDS chunk, so the MFMA delivers S' = q.k - M and
  P = exp2(S') needs ONE v_exp per score; row sums come out of the P.V MFMA (ones row of V^T, accumulator row 72).
  M only moves when a row max exceeds it by more than 2^THR (rare path: rescale O, shift the pending scores,
  rewrite the padding dim).
Body t (starts right after barrier t-1):  4 trailing P.V MFMAs of tile t-1 | QK^T of tile t+1 (20) | P.V of
tile t (20 of 24); beside them: K / V^T fragment reads (4-deep rings), LDS-DMA of K(t+2), V(t+1), exp2 + pack
of tile t, row max of tile t+1.
"""
import argparse
import os

HD = 72
NKS, NDT = 5, 3
KTILE, VTILE = 9216, 12288
KOFF = [0, KTILE]
VOFF = [2 * KTILE, 2 * KTILE + VTILE]
CONST_OFF = 2 * KTILE + 2 * VTILE          # 16-byte chunk {1.0bf16, 0...}: K's padding dims 72..79
SMEM = CONST_OFF + 16
WINDOWS = [4, 25, 24, 40, 26, 43]          # first / last shadow of filler classes A, B, C (see body())
THR_BITS = "0x41000000"                    # 8.0: rescale when a row max exceeds the reference by > 2^8

# ---- physical registers owned by the asm (declared as clobbers by the wrapper)
V_FIRST = 48
SA0, SB0, PB0 = 48, 112, 176
KR0, VR0 = 208, 224                        # fragment rings, 4 slots x 4 registers each
TMP0 = 240                                 # 8 temporaries
MT = [248, 249]                            # row max of the pending score tile, per query block
MM = [250, 251]                            # running reference max M (bf16-exact f32), per query block
TX = [252, 253, 254, 255]
A_O0, A_Q0, A_LAST = 0, 96, 135
S_FIRST, S_LAST = 36, 63
S_KB, S_VB, S_KSTEP, S_KJ, S_VJ = 40, 42, 44, 46, 48
S_TPS, S_NT, S_KDST, S_VDST, S_NKW, S_NVW = 50, 51, 52, 53, 54, 55
S_T, S_KTT, S_VTT, S_TMP, S_HIM, S_KL, S_VL = 56, 57, 58, 59, 60, 62, 63

# ---- asm operands (order = operand numbers in the wrapper's asm statement)
OPERANDS = ["m0out", "m1out",
            "koff0", "koff1", "koff2", "voff0", "voff1", "voff2",
            "fo0", "fo1", "fo2", "fo3", "kc00", "kc01", "kc10", "kc11",
            "kbase", "vbase", "kstep", "kjump", "vjump", "tps", "nt", "kdst", "vdst", "nkw", "nvw"]
OP = {n: "%%%d" % i for i, n in enumerate(OPERANDS)}


def vr(base, n=1):
    return "v%d" % base if n == 1 else "v[%d:%d]" % (base, base + n - 1)


def ar(base, n=1):
    return "a%d" % base if n == 1 else "a[%d:%d]" % (base, base + n - 1)


def S(setbase, u, t2, r=0):
    return setbase + (u * 2 + t2) * 16 + r


def PB(u, g, w=0):
    return PB0 + (u * 4 + g) * 4 + w


def AO(u, d):
    return A_O0 + (u * NDT + d) * 16


def AQ(u, ks):
    return A_Q0 + (u * NKS + ks) * 4


class Stream:
    """instruction list with in-order LDS-read bookkeeping (lgkmcnt) and simple hazard assertions"""

    def __init__(self):
        self.lines = []
        self.pending = []   # tags of outstanding ds_reads, program order
        self.table = []     # (kind, text) for --table

    def emit(self, text, kind="x"):
        ab = getattr(self, "ablate", frozenset()) if getattr(self, "in_body", False) else frozenset()
        if ("noexp" in ab and text.startswith("v_exp")) or ("nobar" in ab and text.startswith("s_barrier")) or \
           ("nodma" in ab and text.startswith("global_load_lds")) or \
           ("nolds" in ab and (text.startswith("ds_read") or "lgkmcnt" in text)) or \
           ("novalu" in ab and kind in ("e", "v") and not text.startswith("v_mfma") and not text.startswith("v_cmp")) or \
           ("nomfma" in ab and text.startswith("v_mfma")) or ("norare" in ab and text.startswith("s_cbranch_vccnz")) or \
           ("nocvt" in ab and text.startswith("v_cvt_pk")) or ("nomax" in ab and (text.startswith("v_max") or text.startswith("v_cmp"))) or \
           ("nowait" in ab and text.startswith("s_waitcnt lgkmcnt")) or ("nosalu" in ab and kind == "s" and not text.startswith("s_add_u32 m0") and "s%d, s%d, 1" % (S_T, S_T) not in text and "s_cmp_lt_u32 s%d, s%d" % (S_T, S_NT) not in text and "_exit" not in text and "_body0" not in text):
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
            after = len(self.pending) - 1 - idx
            self.emit("s_waitcnt lgkmcnt(%d)" % after, "w")
            self.pending = self.pending[idx + 1:]

    def drain(self):
        self.pending = []


# ------------------------------------------------------------------------------------------ pieces
def k_read(st, slot, p, tag):
    """K fragment of pair p = (ks, t2) of the tile in ring slot `slot` -> K ring"""
    ks, t2 = p // 2, p % 2
    dst = KR0 + (p % 4) * 4
    if ks < 4:
        st.ds_read(dst, OP["fo%d" % ks], KOFF[slot] + t2 * 4096, tag)
    else:
        st.ds_read(dst, OP["kc%d%d" % (slot, t2)], 0, tag)


def v_read(st, slot, r, tag):
    g, d = r // NDT, r % NDT
    dst = VR0 + (r % 4) * 4
    st.ds_read(dst, OP["fo%d" % g], VOFF[slot] + d * 4096, tag)


def qk_mfma(st, sn, a):
    ks, t2, u = a // 4, (a // 2) % 2, a % 2
    p = a // 2
    frag = KR0 + (p % 4) * 4
    dst = vr(S(sn, u, t2), 16)
    st.emit("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (dst, vr(frag, 4), ar(AQ(u, ks), 4), "0" if ks == 0 else dst), "M")


def pv_mfma(st, b):
    g, d, u = b // (2 * NDT), (b // 2) % NDT, b % 2
    r = b // 2
    frag = VR0 + (r % 4) * 4
    dst = ar(AO(u, d), 16)
    st.emit("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (dst, vr(frag, 4), vr(PB(u, g), 4), dst), "M")


def exp_group(sc, u, g):
    """exp2 + pack of 8 scores (16 keys of one query block) -> one B-operand fragment of P"""
    t2, r0 = g >> 1, (g & 1) * 8
    ops = []
    for r in range(8):
        x = vr(S(sc, u, t2, r0 + r))
        ops.append(("e", "v_exp_f32 %s, %s" % (x, x)))
    for w in range(4):
        ops.append(("v", "v_cvt_pk_bf16_f32 %s, %s, %s" % (vr(PB(u, g, w)), vr(S(sc, u, t2, r0 + 2 * w)),
                                                           vr(S(sc, u, t2, r0 + 2 * w + 1)))))
    return ops


def rowmax_chain(sn, u, t2, tmp):
    x = lambda r: vr(S(sn, u, t2, r))
    ops = [("v", "v_max3_f32 %s, %s, %s, %s" % (vr(tmp), x(0), x(1), x(2)))]
    for k in range(6):
        ops.append(("v", "v_max3_f32 %s, %s, %s, %s" % (vr(tmp), vr(tmp), x(3 + 2 * k), x(4 + 2 * k))))
    ops.append(("v", "v_max_f32 %s, %s, %s" % (vr(tmp), vr(tmp), x(15))))
    return ops


def lanemax_final():
    """common path: ONE number per lane = max of its 64 pending scores (both query blocks); the exact per-row max
    (other half-wave included) is only formed in the rare path"""
    return [("v", "v_max3_f32 %s, %s, %s, %s" % (vr(MT[0]), vr(TMP0), vr(TMP0 + 1), vr(TMP0 + 2))),
            ("v", "v_max_f32 %s, %s, %s" % (vr(MT[0]), vr(MT[0]), vr(TMP0 + 3))),
            # the compare already here: VCC is old news when the next body's s_cbranch_vccnz reads it
            ("v", "v_cmp_lt_f32 vcc, %s, %s" % (THR_BITS, vr(MT[0])))]


def rowmax_from_chains(st):
    """rare path / prologue: per query block, combine the two chain maxima (TMP0+u: keys 0..31, TMP0+2+u: keys
    32..63 of this lane's half) and the other half-wave's -> MT[u]"""
    for u in range(2):
        ta, tb = TMP0 + u, TMP0 + 2 + u
        st.emit("v_max_f32 %s, %s, %s" % (vr(ta), vr(ta), vr(tb)))
        st.emit("v_mov_b32 %s, %s" % (vr(tb), vr(ta)))
        st.emit("s_nop 1", "n")
        st.emit("v_permlane32_swap_b32 %s, %s" % (vr(ta), vr(tb)))
        st.emit("s_nop 1", "n")
        st.emit("v_max_f32 %s, %s, %s" % (vr(MT[u]), vr(ta), vr(tb)))


def k_dma(st, slot, i, part=3):
    """K loader slot i of this wave -> ring slot `slot` (instruction j = wave + 4 i).  part 1 = M0 write only,
    2 = the DMA only (one other instruction must sit between them), 3 = both with an s_nop"""
    if part & 1:
        st.emit("s_add_u32 m0, s%d, %d" % (S_KDST, KOFF[slot] + 4096 * i), "s")
    if part == 3:
        st.emit("s_nop 0", "n")
    if part & 2:
        st.emit("global_load_lds_dwordx4 %s, s[%d:%d]" % (OP["koff%d" % i], S_KB, S_KB + 1), "g")


def v_dma(st, slot, i, part=3):
    if part & 1:
        st.emit("s_add_u32 m0, s%d, %d" % (S_VDST, (VOFF[slot] - VOFF[0]) + 4096 * i), "s")
    if part == 3:
        st.emit("s_nop 0", "n")
    if part & 2:
        st.emit("global_load_lds_dwordx4 %s, s[%d:%d]" % (OP["voff%d" % i], S_VB, S_VB + 1), "g")


def k_advance(st, uid):
    """point the K loader at its next tile; past the last tile it stays (harmless re-fetch of the last tile)"""
    L = ".L@@_ka%s" % uid
    st.emit("s_add_u32 s%d, s%d, 1" % (S_TMP, S_KL), "s")
    st.emit("s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NT), "s")
    st.emit("s_cbranch_scc0 %s" % L, "s")
    st.emit("s_mov_b32 s%d, s%d" % (S_KL, S_TMP), "s")
    st.emit("s_add_u32 s%d, s%d, s%d" % (S_KB, S_KB, S_KSTEP), "s")
    st.emit("s_addc_u32 s%d, s%d, 0" % (S_KB + 1, S_KB + 1), "s")
    st.emit("s_add_u32 s%d, s%d, 1" % (S_KTT, S_KTT), "s")
    st.emit("s_cmp_lg_u32 s%d, s%d" % (S_KTT, S_TPS), "s")
    st.emit("s_cbranch_scc1 %s" % L, "s")
    st.emit("s_mov_b32 s%d, 0" % S_KTT, "s")
    st.emit("s_add_u32 s%d, s%d, s%d" % (S_KB, S_KB, S_KJ), "s")
    st.emit("s_addc_u32 s%d, s%d, s%d" % (S_KB + 1, S_KB + 1, S_KJ + 1), "s")
    st.label(L)


def v_advance(st, uid):
    L = ".L@@_va%s" % uid
    st.emit("s_add_u32 s%d, s%d, 1" % (S_TMP, S_VL), "s")
    st.emit("s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NT), "s")
    st.emit("s_cbranch_scc0 %s" % L, "s")
    st.emit("s_mov_b32 s%d, s%d" % (S_VL, S_TMP), "s")
    st.emit("s_add_u32 s%d, s%d, 128" % (S_VB, S_VB), "s")
    st.emit("s_addc_u32 s%d, s%d, 0" % (S_VB + 1, S_VB + 1), "s")
    st.emit("s_add_u32 s%d, s%d, 1" % (S_VTT, S_VTT), "s")
    st.emit("s_cmp_lg_u32 s%d, s%d" % (S_VTT, S_TPS), "s")
    st.emit("s_cbranch_scc1 %s" % L, "s")
    st.emit("s_mov_b32 s%d, 0" % S_VTT, "s")
    st.emit("s_add_u32 s%d, s%d, s%d" % (S_VB, S_VB, S_VJ), "s")
    st.emit("s_addc_u32 s%d, s%d, s%d" % (S_VB + 1, S_VB + 1, S_VJ + 1), "s")
    st.label(L)


def k_dma2(st, slot, uid):
    """third K loader slot: only the wave that owns instruction 8 (the 8-dim column image)"""
    L = ".L@@_k2%s" % uid
    st.emit("s_cmp_lt_u32 s%d, 3" % S_NKW, "s")
    st.emit("s_cbranch_scc1 %s" % L, "s")
    k_dma(st, slot, 2)
    st.label(L)


def v_dma2(st, slot, uid):
    L = ".L@@_v2%s" % uid
    st.emit("s_cmp_lt_u32 s%d, 3" % S_NVW, "s")
    st.emit("s_cbranch_scc1 %s" % L, "s")
    v_dma(st, slot, 2)
    st.label(L)


def dma_group(st, which, slot, uid):
    """all LDS-DMA instructions of this wave for one K (or V^T) tile + loader advance (prologue form)"""
    if which == "k":
        k_dma(st, slot, 0)
        k_dma(st, slot, 1)
        k_dma2(st, slot, uid)
        k_advance(st, uid)
    else:
        v_dma(st, slot, 0)
        v_dma(st, slot, 1)
        v_dma2(st, slot, uid)
        v_advance(st, uid)


def fixup(st, sx, uid, init):
    """move the reference max M to (M + max(mt, 0)) [init: to mt], rounded to bf16: shift the pending scores,
    rescale O (not at init: O == 0), rewrite Q's padding dim 72 with -M."""
    if not init:
        st.emit("s_nop 15", "n")
        st.emit("s_nop 15", "n")  # trailing P.V MFMAs -> v_accvgpr_read of O
    rowmax_from_chains(st)
    for u in range(2):
        d, n, pk, f, de, al, t = TX[0], TX[1], TX[2], TX[3], TMP0 + 4, TMP0 + 5, TMP0 + 6
        if init:
            st.emit("v_mov_b32 %s, %s" % (vr(n), vr(MT[u])))
        else:
            st.emit("v_max_f32 %s, 0, %s" % (vr(d), vr(MT[u])))
            st.emit("v_add_f32 %s, %s, %s" % (vr(n), vr(MM[u]), vr(d)))
        st.emit("v_xor_b32 %s, 0x80000000, %s" % (vr(n), vr(n)))              # -(M + mt')
        st.emit("v_cvt_pk_bf16_f32 %s, %s, 0" % (vr(pk), vr(n)))                # lo16 = bf16(-M_new), hi16 = 0
        st.emit("v_lshlrev_b32 %s, 16, %s" % (vr(f), vr(pk)))                   # f32(-M_new)
        st.emit("v_add_f32 %s, %s, %s" % (vr(de), vr(MM[u]), vr(f)))            # M_old - M_new (<= 0 in the loop)
        st.emit("v_xor_b32 %s, 0x80000000, %s" % (vr(MM[u]), vr(f)))            # M = M_new
        if not init:
            st.emit("v_exp_f32 %s, %s" % (vr(al), vr(de)))                      # alpha = 2^(M_old - M_new)
        # Q padding dim 72 lives in lanes 32..63 of word 0 of the k-step-4 fragment
        st.emit("v_accvgpr_read_b32 %s, %s" % (vr(t), ar(AQ(u, 4))))
        st.emit("s_nop 0", "n")
        st.emit("v_cndmask_b32_e64 %s, %s, %s, s[%d:%d]" % (vr(t), vr(t), vr(pk), S_HIM, S_HIM + 1))
        st.emit("s_nop 0", "n")
        st.emit("v_accvgpr_write_b32 %s, %s" % (ar(AQ(u, 4)), vr(t)))
        for t2 in range(2):
            for r in range(16):
                x = vr(S(sx, u, t2, r))
                st.emit("v_add_f32 %s, %s, %s" % (x, x, vr(de)))
        if not init:
            for dd in range(NDT):
                for r0 in range(0, 16, 4):
                    for r in range(r0, r0 + 4):
                        st.emit("v_accvgpr_read_b32 %s, %s" % (vr(TMP0 + (r - r0)), ar(AO(u, dd) + r)))
                    st.emit("s_nop 0", "n")
                    for r in range(r0, r0 + 4):
                        tt = vr(TMP0 + (r - r0))
                        st.emit("v_mul_f32 %s, %s, %s" % (tt, tt, vr(al)))
                    st.emit("s_nop 0", "n")
                    for r in range(r0, r0 + 4):
                        st.emit("v_accvgpr_write_b32 %s, %s" % (ar(AO(u, dd) + r), vr(TMP0 + (r - r0))))
    st.emit("s_nop 7", "n")  # v_accvgpr_write -> MFMA operand


# ------------------------------------------------------------------------------------------ body
def body(st, k, safe, ablate=frozenset()):
    """iteration with ring slot parity k: SC = scores of tile t (k == 0: set A), SN receives tile t+1"""
    sc, sn = (SA0, SB0) if k == 0 else (SB0, SA0)
    cur = k
    uid = "b%d" % k
    st.label(".L@@_body%d" % k)
    # -- top: K fragment reads of pairs 0..3 (tile t+1 sits in ring slot cur^1)
    for p in range(4):
        k_read(st, cur ^ 1, p, ("k", p))
    # -- 4 trailing P.V MFMAs of tile t-1 (fragments read before the barrier) + the K loader's LDS-DMA
    k_dma(st, cur, 0, 1)                 # K(t+2) -> slot cur and V(t+1) -> slot cur^1: one LDS-DMA piece per shadow
    pv_mfma(st, 20)                      # (the M0 write sits before the MFMA: no s_nop needed)
    k_dma(st, cur, 0, 2)
    k_dma(st, cur, 1, 1)
    pv_mfma(st, 21)
    k_dma(st, cur, 1, 2)
    v_dma(st, cur ^ 1, 0, 1)
    pv_mfma(st, 22)
    v_dma(st, cur ^ 1, 0, 2)
    v_dma(st, cur ^ 1, 1, 1)
    pv_mfma(st, 23)
    v_dma(st, cur ^ 1, 1, 2)
    # -- decision: does any row max of tile t exceed the reference by more than 2^THR?
    st.emit("s_cbranch_vccnz .L@@_rare%d" % k)
    st.label(".L@@_entry%d" % k)

    # -- fillers of the 40 shadows behind the QK^T (global MFMA index i = 4..23) and P.V (i = 24..43) MFMAs,
    # paced in CYCLES (v_exp 8, other VALU 4, per-shadow budget ~26 of the MFMA's 32): three classes, each spread
    # uniformly over its window of shadows
    #   A: exp2 + pack of key groups 0..2   shadows  4..25   (P.V of group g starts at i = 24 + 6 g)
    #   B: exp2 + pack of key group 3       shadows 24..40
    #   C: row max of tile t+1              shadows 26..43   (its last QK^T MFMAs are i = 20..23)
    cost = lambda kind_: 8.0 if kind_ == "e" else (2.0 if kind_ == "n" else 4.0)
    clsA, clsB, clsC = [], [], []
    for g in range(4):
        for u in range(2):
            (clsA if g < 3 else clsB).extend(exp_group(sc, u, g))
    ca = [rowmax_chain(sn, 0, 0, TMP0), rowmax_chain(sn, 1, 0, TMP0 + 1)]     # t2 = 0: last MFMAs at i = 20, 21
    cb = [rowmax_chain(sn, 0, 1, TMP0 + 2), rowmax_chain(sn, 1, 1, TMP0 + 3)]  # t2 = 1: last MFMAs at i = 22, 23
    clsC.extend([o for pair in zip(*ca) for o in pair])
    n_first_b = len(clsC)                # chain-B ops (t2 = 1) must not start before shadow 28
    clsC.extend([o for pair in zip(*cb) for o in pair])
    clsC.extend(lanemax_final())
    W = WINDOWS
    classes = [[clsA, W[0], W[1], 0, 0.0], [clsB, W[2], W[3], 0, 0.0], [clsC, W[4], W[5], 0, 0.0]]  # items, first, last, next, emitted
    totals = [sum(cost(o[0]) for o in c[0]) for c in classes]
    mf = [("qk", a) for a in range(20)] + [("pv", b) for b in range(20)]
    for n, (kind, idx) in enumerate(mf):
        i = 4 + n
        if kind == "qk":
            if idx % 4 == 0:             # pairs 2m and 2m+1 with one wait (both were issued >= 3 pairs ago)
                st.need(("k", idx // 2 + 1))
            qk_mfma(st, sn, idx)
        else:
            if idx % 4 == 0:
                st.need(("v", idx // 2 + 1))
            pv_mfma(st, idx)
        if safe:
            st.emit("s_nop 7", "n")
        used = 0.0
        # ring reads: after the 2nd MFMA of pair x its ring slot is free -> read pair x + 4
        if kind == "qk" and idx % 2 == 1:
            p = idx // 2
            if p + 4 < 10:
                k_read(st, cur ^ 1, p + 4, ("k", p + 4))
            else:
                v_read(st, cur, p + 4 - 10, ("v", p + 4 - 10))   # V pairs 0..3 behind the last K pairs
            used += 4
        if kind == "pv" and idx % 2 == 1:
            r = idx // 2
            if r + 4 < 12:
                v_read(st, cur, r + 4, ("v", r + 4))
                used += 4
        if i == 4:
            k_dma2(st, cur, uid); used += 12
        elif i == 5:
            v_dma2(st, cur ^ 1, uid); used += 12
        elif i == 6:
            k_advance(st, uid); used += 28
        elif i == 7:
            v_advance(st, uid); used += 28
        for ci, c in enumerate(classes):
            items, first, last, nx, em = c
            if i < first:
                continue
            frac = min(1.0, (i - first + 1) / float(last - first + 1))
            while c[3] < len(items):
                kind_, text = items[c[3]]
                if ci == 2 and c[3] >= n_first_b and i < max(28, W[4]):
                    break
                behind = totals[ci] * frac - c[4]
                if behind <= 0 and i < last:
                    break
                if used >= 30 and i < last:
                    break
                st.emit(text, kind_)
                used += cost(kind_)
                c[4] += cost(kind_)
                c[3] += 1
    for c in classes:
        assert c[3] == len(c[0]), "unscheduled filler work"
    # -- end of body: DMA landed + every fragment read retired, then the tile barrier
    st.emit("s_waitcnt vmcnt(0) lgkmcnt(0)", "w")
    st.drain()
    st.emit("s_barrier", "B")
    st.emit("s_add_u32 s%d, s%d, 1" % (S_T, S_T), "s")
    st.emit("s_cmp_lt_u32 s%d, s%d" % (S_T, S_NT), "s")
    st.emit("s_cbranch_scc0 .L@@_exit", "s")
    if k == 1:
        st.emit("s_branch .L@@_body0", "s")


def generate(safe=False, ablate=frozenset()):
    st = Stream()
    st.ablate = ablate
    e = st.emit
    # ---- copy the mutable scalars into asm-owned SGPRs
    e("s_mov_b64 s[%d:%d], %s" % (S_KB, S_KB + 1, OP["kbase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_VB, S_VB + 1, OP["vbase"]))
    e("s_mov_b32 s%d, %s" % (S_KSTEP, OP["kstep"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_KJ, S_KJ + 1, OP["kjump"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_VJ, S_VJ + 1, OP["vjump"]))
    e("s_mov_b32 s%d, %s" % (S_TPS, OP["tps"]))
    e("s_mov_b32 s%d, %s" % (S_NT, OP["nt"]))
    e("s_mov_b32 s%d, %s" % (S_KDST, OP["kdst"]))
    e("s_mov_b32 s%d, %s" % (S_VDST, OP["vdst"]))
    e("s_mov_b32 s%d, %s" % (S_NKW, OP["nkw"]))
    e("s_mov_b32 s%d, %s" % (S_NVW, OP["nvw"]))
    e("s_mov_b32 s%d, 0" % S_T)
    e("s_mov_b32 s%d, 0" % S_KTT)
    e("s_mov_b32 s%d, 0" % S_VTT)
    e("s_mov_b32 s%d, 0" % S_KL)
    e("s_mov_b32 s%d, 0" % S_VL)
    e("s_mov_b32 s%d, 0" % S_HIM)
    e("s_mov_b32 s%d, -1" % (S_HIM + 1))
    for u in range(2):
        e("v_mov_b32 %s, 0" % vr(MM[u]))
    for r in range(A_O0, A_Q0):
        e("v_accvgpr_write_b32 %s, 0" % ar(r))
    # ---- prologue: K0, V0 -> slot 0, K1 -> slot 1
    dma_group(st, "k", 0, "p0")
    dma_group(st, "v", 0, "p1")
    dma_group(st, "k", 1, "p2")
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    # scores of tile 0 -> set A (Q's padding dim is 0: raw scores)
    for p in range(10):
        k_read(st, 0, p, ("k", p))
        st.need(("k", p))
        qk_mfma(st, SA0, 2 * p)
        qk_mfma(st, SA0, 2 * p + 1)
    e("s_barrier")                      # every wave has read K0: slot 0 may be refilled
    e("s_nop 15")
    e("s_nop 15")
    for op in (rowmax_chain(SA0, 0, 0, TMP0) + rowmax_chain(SA0, 1, 0, TMP0 + 1) + rowmax_chain(SA0, 0, 1, TMP0 + 2)
               + rowmax_chain(SA0, 1, 1, TMP0 + 3)):
        e(op[1])
    fixup(st, SA0, "init", init=True)
    # what body 0 does before its entry point: first LDS-DMA pieces of K(2) -> slot 0 and V(1) -> slot 1, first K
    # fragment reads of tile 1
    k_dma(st, 0, 0)
    k_dma(st, 0, 1)
    v_dma(st, 1, 0)
    v_dma(st, 1, 1)
    for p in range(4):
        k_read(st, 1, p, ("k", p))
    e("s_branch .L@@_entry0")
    pend = list(st.pending)
    # ---- the two loop bodies
    st.pending = []
    st.in_body = True
    body(st, 0, safe, ablate)
    st.pending = []
    body(st, 1, safe, ablate)
    st.in_body = False
    # ---- rare paths
    for k in range(2):
        st.label(".L@@_rare%d" % k)
        fixup(st, SA0 if k == 0 else SB0, "r%d" % k, init=False)
        e("s_branch .L@@_entry%d" % k)
    # ---- exit: the 4 trailing P.V MFMAs of the last tile
    st.label(".L@@_exit")
    for b in range(20, 24):
        pv_mfma(st, b)
    e("s_nop 15")
    e("s_nop 15")
    e("v_mov_b32 %s, %s" % (OP["m0out"], vr(MM[0])))
    e("v_mov_b32 %s, %s" % (OP["m1out"], vr(MM[1])))
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table", action="store_true")
    ap.add_argument("--exp", nargs="*", default=["safe"], help="experimental variants 1..3: safe | ablations joined by + (noexp nobar nodma nolds novalu nomfma): timing only, wrong results")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "open_sora_amd", "csrc"))
    args = ap.parse_args()
    variants = [(False, frozenset())] + [(x == "safe", frozenset(x.split("+")) - {"safe"}) for x in args.exp]
    while len(variants) < 4:
        variants.append((True, frozenset()))
    for vi, (safe, ablate) in enumerate(variants[:4]):
        st = generate(safe, ablate)
        if args.table and vi == 0:
            gap = []
            for kind, text in st.table:
                if kind == "M":
                    print("".join(gap)); gap = ["M "]
                elif kind == "L":
                    print("".join(gap)); gap = []; print(text + ":")
                else:
                    gap.append({"x": "v", "w": "w"}.get(kind, kind))
            print("".join(gap))
        with open(os.path.join(args.out, "attention_asm72_body_v%d.inc" % vi), "w") as f:
            f.write("// GENERATED by tools/gen_attn_asm.py -- do not edit.  variant %d: %s\n" %
                    (vi, "production" if vi == 0 else ("hazard-padded (debug)" if safe else "timing ablation " + "+".join(sorted(ablate)))))
            for ln in st.lines:
                f.write('"%s\\n"\n' % ln.replace("@@", "osk72v%d" % vi))
    # register / operand contract for the wrapper
    with open(os.path.join(args.out, "attention_asm72_regs.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_attn_asm.py -- do not edit.\n")
        f.write("#define OSK72_SMEM %d\n#define OSK72_CONST_OFF %d\n" % (SMEM, CONST_OFF))
        f.write("#define OSK72_KTILE %d\n#define OSK72_VTILE %d\n#define OSK72_VOFF0 %d\n" % (KTILE, VTILE, VOFF[0]))
        f.write("#define OSK72_AQ0 %d\n" % A_Q0)
        clob = ['"v%d"' % i for i in range(V_FIRST, 256)] + ['"a%d"' % i for i in range(0, A_LAST + 1)] + \
               ['"s%d"' % i for i in range(S_FIRST, S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']
        f.write("#define OSK72_CLOBBERS %s\n" % ", ".join(clob))
        f.write("#define OSK72_A_CLOBBERS %s\n" % ", ".join('"a%d"' % i for i in range(0, A_LAST + 1)))
        for u in range(2):   # Q fragment words of query block u (operands %0..%19) -> AGPRs
            f.write("#define OSK72_QW%d %s\n" % (u, " ".join('"v_accvgpr_write_b32 a%d, %%%d\\n"' % (AQ(u, 0) + i, i) for i in range(NKS * 4))))
        for u in range(2):   # O^T row tile (u, d) -> operands %0..%15
            for d in range(NDT):
                f.write("#define OSK72_OR%d %s\n" % (u * NDT + d, " ".join('"v_accvgpr_read_b32 %%%d, a%d\\n"' % (i, AO(u, d) + i) for i in range(16))))


if __name__ == "__main__":
    main()
