#!/usr/bin/env python
"""Generator of the hand-scheduled gfx950 main loops of the flash-attention kernels for head_dim 72
(open_sora_amd/csrc/attention_asm72.hip includes the emitted attention_asm72_n{NU}_v{VAR}.inc) and head_dim 128
(attention_asm128.hip, attention_asm128_n2_v{VAR}.inc); class Geometry holds what differs between the two.

Why a generator: the MFMA shadow (32 cycles, ~5 issue slots) has to be filled by hand; hipcc's scheduler clusters
the softmax VALU work behind the MFMAs and shuffles accumulators between the VGPR and AGPR halves
(tools/isa_stream.py on the compiler-scheduled twin of rounds 1-2, attention_w64.hip, showed it).  The schedule below is explicit and reproducible;
`python tools/gen_attn_asm.py --table NU [--hd 128]` prints it shadow by shadow.

Two layouts of the same dataflow, workgroup = 256 query rows, KV tile = 64 keys:
  NU = 2: 4 waves x 64 rows (two 32-row query blocks u per wave share every K / V^T fragment), one wave per SIMD,
          392 registers per wave;
  NU = 1: 8 waves x 32 rows, two waves per SIMD (<= 256 registers each): twice the LDS fragment traffic, but each
          wave's fillers also sit in the shadow of its partner's MFMAs.
Dataflow (operand conventions of rounds 1-2's attention_w64.hip, the compiler-scheduled twin that validated the LDS images and
the pipeline on the GPU): S^T = K . Q^T and O^T += V^T . P^T on v_mfma_f32_32x32x16_bf16 with swapped operands (a
lane owns one query).  Q is pre-multiplied by scale*log2(e); the reference max M (bf16-exact, log2 units) sits,
negated, in Q's padding dim 72 and K's padding dim 72 reads 1.0 from a constant LDS chunk, so the MFMA delivers
S' = q.k - M and P = exp2(S') needs ONE v_exp per score; row sums come out of the P.V MFMA (ones row of V^T,
accumulator row 72).  M only moves when some score exceeds it by more than 2^THR (rare path: rescale O, shift the
pending scores, rewrite the padding dim).
FAST bodies (`generate(..., fast=True)`, attention_asm*_n2_f0.inc): the loop the single-GPU denoise step runs.  The gap of a
32-cycle MFMA hides at most ~5 other instructions of the one wave on the SIMD (MI355X_MICROARCH.md); the general body carries
~230 per 44 MFMAs, i.e. it is ISSUE-bound.  Two launch-uniform facts remove a third of them:
  * a caller-supplied BOUND on the scores (|q.k| <= B, log2 units; the model path knows it from the QK-norm scale vectors): the
    reference max is the constant B -- Q's padding dim holds -B from the start, P = exp2(s - B) <= 1 can never overflow -- so
    the per-lane max chains (34 VALU per tile), the compare, the branch and the whole rare path disappear;
  * segment ends are EVENTS, not per-tile state: both loaders advance with six branch-free SALU instructions (clamped at the last
    tile) behind one compare + (not taken) branch against the tile index of the loader's next event; the out-of-line event
    code (after the loop, one copy per advance site) does what the general body does on every tile -- the K loader's per-lane
    offsets switch to the clamped set (KCUR registers) while it sits on a ragged segment-last tile, the segment jump, the V^T
    ring slot's ones row becoming that tile's key-validity mask and ones again two tiles later.  Round 4: this made the
    bounded body usable for every sequence-parallel call (n_seg = P) and for ragged key counts (the reference's 256 px shape
    L = 8,828 = 137 x 64 + 60); round 3's FAST body took one segment of whole tiles only.
Body t (starts right after barrier t-1): the last 2 fragment pairs' P.V MFMAs of tile t-1 | QK^T of tile t+1 |
P.V of tile t (first 10 of 12 fragment pairs); beside them: K / V^T fragment reads (4-deep rings), LDS-DMA of
K(t+2), V(t+1), exp2 + pack of tile t, max of tile t+1.
"""
import argparse
import os

THR_BITS = "0x41000000"                    # 8.0: move M when a score exceeds the reference by > 2^8


class Geometry:
    """head-dim dependent tile geometry.
    hd 72 : 4 real QK^T k-steps + 1 half-real one (dims 64..79: 8 real dims from the column image, the padding half
            carries M), 3 O^T row tiles (72 real rows, row 72 = ones row).
    hd 128: 8 real k-steps + 1 pure padding step whose K fragment is a CONSTANT register quad {1.0, 0...} (no LDS
            read) and whose Q fragment carries M; 4 real O^T row tiles + a 5th that only holds the ones row 128."""

    def __init__(self, hd, pv8=False, pv16=False, nom=False):
        self.HD = hd
        # nom (head_dim 128 FAST body): no padding k-step at all.  That step exists only to carry the reference max through the
        # MFMA; with a score bound B <= 56 no reference is needed for range control -- P = exp2(s), |s| <= B, stays inside
        # [2^-56, 2^56] and every sum inside f32 -- so QK^T is 8 k-steps instead of 9 (72 instead of 76 MFMAs per tile)
        self.NOM = nom
        assert not nom or (hd == 128 and not pv8)
        self.PV8 = pv8   # P.V on the fp8 MFMA (v_mfma_f32_32x32x64_f8f6f4): P and V^T as OCP e4m3, see the header
        # P.V on v_mfma_f32_16x16x32_bf16 (head_dim 72): O^T in 16-row blocks, so the padding of the 72 dims + ones row shrinks
        # from 96 rows (3 x 32) to 80 (5 x 16): 40 x 16 cycles instead of 24 x 32 per 64-key tile and 64 query rows.  The scores
        # still come out of 32 x 32 x 16 MFMAs (QK^T at 80 padded dims; 16x16x32 would pad it to 96): a lane's 16 P values of a
        # 32-key half (one query, keys 8 j + 4 (lane / 32) + i) become two 16x16x32 B operands (queries 0..15 / 16..31 of the
        # block, 8 keys per lane) by ONE v_permlane16_swap per packed register pair; V^T's key order inside a 32-key half is
        # baked to match by osk_v_transpose_bf16 (chunk 4 t2 + r of a 64-key row = the keys of lane row r: PV16_KEYS).
        self.PV16 = pv16
        assert not (pv8 and pv16) and (not pv16 or hd == 72)
        if hd == 72:
            self.NKS, self.NDT, self.KIMG, self.HAS_COL = 5, 3, 1, True
            self.M_KS, self.M_HI = 4, 1          # M sits in k-step 4, lanes 32..63 (dims 72..79), word 0
            self.NKD = self.NVD = 9
        elif hd == 128:
            self.NKS, self.NDT, self.KIMG, self.HAS_COL = 9, 5, 2, False
            self.M_KS, self.M_HI = 8, 0          # k-step 8 (dims 128..143), lanes 0..31 (dims 128..135), word 0
            self.NKD = self.NVD = 16
        else:
            raise ValueError(hd)
        self.NPK = 2 * (self.NKS - (1 if nom else 0))   # QK^T fragment pairs (k-step, 32-key half)
        self.NPK_READ = 2 * (self.NKS - (0 if self.HAS_COL else 1))   # ... that are read from LDS
        self.KTILE = self.KIMG * 8192 + (1024 if self.HAS_COL else 0)
        if pv8:
            # V^T tile in e4m3: 64-byte rows (64 keys), ONE fp8 MFMA (K = 64) per O^T row tile.  The global tensor
            # carries RP rows per head: the hd dims, the baked ones / key-validity row hd, zero rows up to a multiple
            # of 16 (one LDS-DMA instruction moves 16 rows), so the kernel maintains no ones row itself.
            self.NPV = self.NDT                  # P.V "pairs" = O^T row tiles
            self.RP = (hd + 1 + 15) // 16 * 16
            self.NVD = self.RP // 16
            self.VROW = 64
            self.VTILE = max(self.RP, self.NDT * 32 if self.HAS_COL else self.RP) * 64
            self.VSTEP = 64                      # bytes per 64-key tile along a V^T row
        elif pv16:
            self.NDB = 5                         # O^T row blocks of 16: dims 0..71, ones row 72, zero rows 73..79
            self.NPV = 2 * self.NDB              # P.V fragment "pairs" (32-key half, row block): 2 NU MFMAs each
            self.VROW = 128
            self.VSTEP = 128
            self.VTILE = self.NDB * 16 * 128
        else:
            self.NPV = 4 * self.NDT              # P.V fragment pairs (key group, row tile)
            self.VROW = 128
            self.VSTEP = 128
            self.VTILE = self.NDT * 32 * 128 if self.HAS_COL else (hd + 2) * 128
        if self.HAS_COL:
            self.KOFF = [0, self.KTILE]
            self.VOFF = [2 * self.KTILE, 2 * self.KTILE + self.VTILE]
        else:
            # hd 128: ds_read immediates are 16 bits, so the tiles have to end below 64 KB + one fragment: the V^T
            # tile keeps only rows 0..129 (128 dims, the ones row, one zero row); the last row tile's fragment reads
            # run into the NEXT region -- garbage that only reaches accumulator rows 129..159, which nobody reads --
            # and V^T sits in front of K so that the largest immediate is K slot 1's (61952)
            self.VOFF = [0, self.VTILE]
            self.KOFF = [2 * self.VTILE, 2 * self.VTILE + self.KTILE]
        # hd 72: 16-byte chunk {1.0bf16, 0...} = K's padding dims 72..79, one copy per K ring slot, KTILE apart
        self.CONST_OFF = 2 * self.KTILE + 2 * self.VTILE
        self.SMEM = self.CONST_OFF + (self.KTILE + 16 if self.HAS_COL else 0)

# key order of V^T inside a 32-key half for the 16x16x32 P.V product: the 16-byte chunk r (= lane row r of the A / B operands) holds
PV16_KEYS = [[0, 1, 2, 3, 8, 9, 10, 11], [16, 17, 18, 19, 24, 25, 26, 27], [4, 5, 6, 7, 12, 13, 14, 15], [20, 21, 22, 23, 28, 29, 30, 31]]

# ---- asm-owned SGPRs
S_FIRST, S_LAST = 36, 67
S_HI2 = 66                                 # lanes 32..63 (hd 128, where S_HIM selects lanes 0..31)
S_KRG, S_VRG, S_FLG, S_NRG = 36, 38, 45, 64  # lane masks: K / V^T loader sits on a ragged (segment-last) tile; flags
# (S_FLG bit 1: this wave maintains the ones rows; S_NRG = 1 when the launch has NO ragged tile)
S_KB, S_VB, S_KSTEP, S_KJ, S_VJ = 40, 42, 44, 46, 48
S_TPS, S_NT, S_KDST, S_VDST, S_NKW, S_NVW = 50, 51, 52, 53, 54, 55
S_T, S_KTT, S_VTT, S_TMP, S_HIM, S_KL, S_VL = 56, 57, 58, 59, 60, 62, 63

# FAST bodies: tile index (loader numbering) at which the loader's next segment-end event is due / of the current segment's last
# tile (S_KRG / S_VRG / S_KTT / S_VTT belong to the general body's per-tile bookkeeping and are dead in the FAST bodies)
S_KNEXT, S_VNEXT, S_KE, S_VE = S_KTT, S_VTT, S_KRG, S_KRG + 1


def operand_names(geo, nslot_k, nslot_v):
    """asm operands (order = operand numbers in the wrapper's asm statement); at most 30"""
    names = ["m0out", "m1out"] + ["koff%d" % i for i in range(nslot_k)] + ["voff%d" % i for i in range(nslot_v)] + \
            ["fo0", "fo1", "fo2", "fo3"] + (["kc0", "kc1"] if geo.HAS_COL else []) + \
            ["koffL%d" % i for i in range(nslot_k)] + \
            (["vf0", "vf1"] if geo.PV8 else ["maskval", "onesaddr"]) + \
            (["vo0", "vo1"] if geo.PV16 else []) + \
            ["kbase", "vbase", "kstep", "kjump", "vjump"] + (["tpsnt"] if geo.PV16 else ["tps", "nt"]) + ["kdst", "vdst"]
    # the K loader's conditional last slot only exists when 4 | NKD does not hold; otherwise no slot count is needed
    # (PV16: the two V^T fragment addresses took two operand slots, so tps | nt << 16 and nvw | nkw << 16 travel packed)
    names += ["nkvw"] if geo.PV16 else ((["nkw"] if geo.NKD % 4 or geo.NKD % 8 else []) + ["nvw"])
    assert len(names) <= 30, len(names)
    return names


def vr(base, n=1):
    return "v%d" % base if n == 1 else "v[%d:%d]" % (base, base + n - 1)


def ar(base, n=1):
    return "a%d" % base if n == 1 else "a[%d:%d]" % (base, base + n - 1)


N2_BUDGET = {"m32": 24, "pv16": 9, "pv8": 60}   # filler cycles per MFMA shadow of the 256-row bodies (round 5: 24 / 9 instead of 30 / 13, as in
# the wide body -- the MFMA's own issue takes 4.4 cycles of its shadow; --n2-budget: experiments)
FAST_WINDOWS_OVERRIDE = {}   # (hd, nu) -> [a0, a1, b0, b1, c0, c1]; set by --fast-windows (experiments) or below (production)


class Layout:
    """register file and schedule geometry for NU query blocks per wave"""

    def __init__(self, nu, hd=72, pv8=False, pv16=False, nom=False):
        self.NU = nu
        self.G = G = Geometry(hd, pv8, pv16, nom)
        self.PV8 = pv8
        self.PV16 = pv16
        self.MPP = 2 * nu if pv16 else nu      # MFMAs per P.V pair (pv16: query blocks of 16: two per 32-row block u)
        NKS, NDT = G.NKS, G.NDT
        self.NW = 8 // nu                      # waves per workgroup
        self.NSLOT = (G.NKD + self.NW - 1) // self.NW   # K LDS-DMA slots per wave and tile
        self.LAST_COND = G.NKD % self.NW != 0  # the last slot only exists on some waves
        self.NSLOT_V = (G.NVD + self.NW - 1) // self.NW
        self.LAST_COND_V = G.NVD % self.NW != 0
        nk_ops = 3 if hd == 72 else self.NSLOT  # hd 72: one wrapper operand list for both layouts
        self.OPERANDS = operand_names(G, nk_ops, self.NSLOT_V if pv8 else nk_ops)
        self.OP = {n: "%%%d" % i for i, n in enumerate(self.OPERANDS)}
        self.RD = 2 if pv8 else 4              # depth of the V^T fragment ring, in P.V pairs
        self.NTP = 1 if pv8 else 2             # P.V pairs whose MFMAs trail across the tile barrier
        self.V_FIRST = 48 if G.HAS_COL else 44
        self.KC0 = None if G.HAS_COL else 44   # constant K fragment of the padding k-step: 4 registers
        self.SA0 = 48
        self.SB0 = self.SA0 + 32 * nu
        self.PB0 = self.SB0 + 32 * nu
        self.KR0 = self.PB0 + (8 if pv8 else 16) * nu   # fragment rings, 16 registers each (pv8: V^T 2 slots x 8)
        self.VR0 = self.KR0 + 16
        self.TMP0 = self.VR0 + 16              # 8 temporaries
        self.KCUR = self.TMP0                  # FAST bodies (no max chains): the K loader's per-lane offsets in force, one per slot
        self.MT = [self.TMP0 + 8, self.TMP0 + 9]
        self.MM = [self.TMP0 + 10, self.TMP0 + 11]
        self.TX = [self.TMP0 + 12 + i for i in range(4)]
        self.V_END = self.TMP0 + 16
        self.A_O0 = 0
        self.A_Q0 = (4 * G.NDB * 2 * nu) if pv16 else 16 * NDT * nu
        self.A_END = self.A_Q0 + 4 * NKS * nu
        self.NTRAIL = self.NTP * self.MPP      # MFMAs of the trailing P.V pairs
        self.NQK = G.NPK * nu
        self.NPVB = (G.NPV - self.NTP) * self.MPP    # P.V MFMAs inside the body
        self.I_QK0 = self.NTRAIL               # global shadow index of the first QK^T MFMA
        self.I_PV0 = self.I_QK0 + self.NQK
        last = self.I_PV0 + self.NPVB - 1
        # first / last shadow of filler classes A (exp2+pack of key groups 0..2), B (group 3: before the P.V MFMAs of
        # key group 3, which start at pair 3 NDT), C (max of tile t+1: after its last QK^T MFMAs)
        if pv8:
            # every P fragment of the tile has to be packed before the FIRST P.V MFMA (it consumes all 64 keys): exp2 +
            # pack of all four key groups sit under the QK^T MFMAs of the next tile, the max of that tile under the
            # (64-cycle) P.V MFMAs
            self.WINDOWS = [self.I_QK0, self.I_PV0 - 3, self.I_QK0 + self.NQK // 2, self.I_PV0 - 2, self.I_PV0 + 1, last]
            self.C_T2_RELEASE = self.I_PV0 + 2
        elif pv16:
            # shadows: 8 trailing P.V (16 cycles) | 20 QK^T (32 cycles) from I_QK0 = 8 | 32 P.V from I_PV0 = 28; class A = exp2 +
            # pack + lane-row swaps of keys 0..31 (before the first P.V MFMA, shadow 28), B = keys 32..63 (before pair 5,
            # shadow 48), C = max of tile t+1 (after its QK^T MFMAs)
            assert nu == 2
            self.WINDOWS = [1, 26, 20, 46, 29, last]
            self.C_T2_RELEASE = 31
        elif hd == 72:
            self.WINDOWS = [4, 25, 24, 40, 26, 43] if nu == 2 else [2, 13, 12, 19, 14, 21]
            self.C_T2_RELEASE = 28 if nu == 2 else 15   # chains over keys 32..63: their last QK^T MFMAs come last
        else:
            self.WINDOWS = [self.I_QK0, self.I_PV0 + 1, self.I_PV0, self.I_PV0 + 3 * NDT * nu - 2, self.I_PV0 + 2, last]
            self.C_T2_RELEASE = self.I_PV0 + 2 * nu
        self.S_ONESMASK = S_HIM if G.M_HI else S_HI2   # lanes 32..63
        # FAST bodies carry no class C: exp2 + pack paced evenly over the whole iteration (deadlines: group g before the P.V
        # pairs of group g, which start at shadow I_PV0 + g NDT NU)
        self.FAST_WINDOWS = FAST_WINDOWS_OVERRIDE.get((hd, nu)) or self.WINDOWS

    def S(self, setbase, u, t2, r=0):
        return setbase + (u * 2 + t2) * 16 + r

    def PB(self, u, g, w=0):
        """bf16: P fragment of key group g (4 registers).  pv8: register w (0..7) of the one 32-byte fragment (g = 0)"""
        return self.PB0 + u * 8 + w if self.PV8 else self.PB0 + (u * 4 + g) * 4 + w

    def AO(self, u, d):
        return self.A_O0 + (u * self.G.NDT + d) * 16

    def AO16(self, u, qb, db):
        """pv16: the 4 accumulator registers of O^T rows 16 db .. + 15 x queries 16 qb .. + 15 of block u (lane: query l % 16,
        rows 4 (l / 16) + i)"""
        return self.A_O0 + ((u * 2 + qb) * self.G.NDB + db) * 4

    def AQ(self, u, ks):
        return self.A_Q0 + (u * self.G.NKS + ks) * 4


class Stream:
    """instruction list with in-order LDS-read bookkeeping (lgkmcnt)"""

    def __init__(self, ablate=frozenset()):
        self.lines, self.pending, self.table = [], [], []
        self.ablate, self.in_body = ablate, False
        self.events = []   # FAST bodies: (which, uid, ring slot, V^T step) of every advance site, for fast_events()

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
def k_frag(L, p):
    """register quad that holds the K fragment of pair p: a ring slot, or the constant padding fragment (hd 128)"""
    return L.KR0 + (p % 4) * 4 if p < L.G.NPK_READ else L.KC0


def k_read(st, L, slot, p, tag):
    """K fragment of pair p = (ks, t2) of the tile in ring slot `slot` -> K ring"""
    ks, t2 = p // 2, p % 2
    assert p < L.G.NPK_READ
    dst = L.KR0 + (p % 4) * 4
    if ks < 4 * L.G.KIMG:
        st.ds_read(dst, L.OP["fo%d" % (ks % 4)], L.G.KOFF[slot] + (ks // 4) * 8192 + t2 * 4096, tag)
    else:
        st.ds_read(dst, L.OP["kc%d" % t2], L.G.KOFF[slot], tag)


def v_read(st, L, slot, r, tag):
    if L.PV8:   # row tile r: 32 rows of 64 bytes; a lane's 32-byte half row = two swizzled 16-byte chunks
        base = L.VR0 + (r % 2) * 8
        st.ds_read(base, L.OP["vf0"], L.G.VOFF[slot] + r * 2048, ("vx", tag[1]))
        st.ds_read(base + 4, L.OP["vf1"], L.G.VOFF[slot] + r * 2048, tag)
        return
    if L.PV16:   # pair r = (32-key half t2, row block db): lane (row l / 16, dim l % 16) reads chunk 4 t2 + l / 16 of its row
        t2, db = r // L.G.NDB, r % L.G.NDB
        st.ds_read(L.VR0 + (r % 4) * 4, L.OP["vo%d" % t2], L.G.VOFF[slot] + db * 2048, tag)
        return
    g, d = r // L.G.NDT, r % L.G.NDT
    st.ds_read(L.VR0 + (r % 4) * 4, L.OP["fo%d" % g], L.G.VOFF[slot] + d * 4096, tag)


def qk_mfma(st, L, sn, a):
    p, u = a // L.NU, a % L.NU
    ks, t2 = p // 2, p % 2
    dst = vr(L.S(sn, u, t2), 16)
    st.emit("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (dst, vr(k_frag(L, p), 4), ar(L.AQ(u, ks), 4),
                                                       "0" if ks == 0 else dst), "M")


def pv_mfma(st, L, b):
    if L.PV16:   # O^T[16 dims x 16 queries] += V^T[16 x 32 keys] . P^T[32 keys x 16 queries]
        r, sub = b // L.MPP, b % L.MPP
        u, qb = sub // 2, sub % 2
        t2, db = r // L.G.NDB, r % L.G.NDB
        dst = ar(L.AO16(u, qb, db), 4)
        st.emit("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (dst, vr(L.VR0 + (r % 4) * 4, 4), vr(L.PB(u, 2 * t2 + qb), 4), dst), "X")
        return
    r, u = b // L.NU, b % L.NU
    if L.PV8:   # one K = 64 fp8 MFMA per O^T row tile: all 64 keys of the tile at once
        dst = ar(L.AO(u, r), 16)
        st.emit("v_mfma_f32_32x32x64_f8f6f4 %s, %s, %s, %s" % (dst, vr(L.VR0 + (r % 2) * 8, 8), vr(L.PB(u, 0), 8), dst), "F")
        return
    g, d = r // L.G.NDT, r % L.G.NDT
    if "pv80" in st.ablate and st.in_body and L.G.HD == 72 and d == 2 and g % 2 == 1:
        return   # timing ablation: the matrix time of an 80-row P.V product (5 x 16 rows) instead of 96 (3 x 32)
    dst = ar(L.AO(u, d), 16)
    st.emit("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (dst, vr(L.VR0 + (r % 4) * 4, 4), vr(L.PB(u, g), 4), dst), "M")


def exp_group(L, sc, u, g):
    """exp2 + pack of 8 scores (16 keys of one query block) -> one B-operand fragment of P"""
    t2, r0 = g >> 1, (g & 1) * 8
    ops = []
    for r in range(8):
        x = vr(L.S(sc, u, t2, r0 + r))
        ops.append(("e", "v_exp_f32 %s, %s" % (x, x)))
    if L.PV8:
        # e4m3 bytes: dword t2 * 4 + j (j = accumulator register / 4) holds keys 32 t2 + 8 j + 4 (lane / 32) + 0..3 --
        # the byte order osk_v_transpose_fp8 bakes into the V^T rows
        for w2 in range(2):
            dst = vr(L.PB(u, 0, t2 * 4 + (g & 1) * 2 + w2))
            x = lambda i: vr(L.S(sc, u, t2, r0 + 4 * w2 + i))
            ops.append(("v", "v_cvt_pk_fp8_f32 %s, %s, %s" % (dst, x(0), x(1))))
            ops.append(("v", "v_cvt_pk_fp8_f32 %s, %s, %s op_sel:[0,0,1]" % (dst, x(2), x(3))))
        return ops
    for w in range(4):
        ops.append(("v", "v_cvt_pk_bf16_f32 %s, %s, %s" % (vr(L.PB(u, g, w)), vr(L.S(sc, u, t2, r0 + 2 * w)),
                                                           vr(L.S(sc, u, t2, r0 + 2 * w + 1)))))
    return ops


def swap_group(L, u, t2, guard=True):
    """pv16: the 8 packed P registers of one 32-key half (lane = one query of the 32-row block, 16 keys) -> two 16x16x32 B operands
    (queries 0..15 in PB(u, 2 t2), queries 16..31 in PB(u, 2 t2 + 1); lane row r holds the keys PV16_KEYS[r]): v_permlane16_swap
    exchanges the odd lane rows of the first register with the even rows of the second.  s_nop: VALU write -> permlane read,
    permlane write -> MFMA operand read."""
    # guard = False (the wide body): no nops -- the order of the fillers keeps the packs >= 3 instructions in front of the swap and
    # the swap >= 3 in front of its MFMA, which _check_swap_distance() verifies on the emitted stream (in-step A/B: -1.5 .. -2.6 % per launch)
    nop = [] if (NO_SWAP_NOPS or not guard) else [("n", "s_nop 1")]
    ops = list(nop)
    for i in range(4):
        ops.append(("v", "v_permlane16_swap_b32 %s, %s" % (vr(L.PB(u, 2 * t2, i)), vr(L.PB(u, 2 * t2 + 1, i)))))
    return ops + nop


NO_SWAP_NOPS = False   # experiment (--wide-exp noswapnop): rely on the distance the filler order gives (checked by _check_swap_distance)


def _check_swap_distance(st):
    """without the guard nops: >= 2 instructions between the v_cvt_pk that writes a P register and the lane-row swap that reads it,
    and between a swap and the MFMA that reads its registers"""
    import re as _re
    last_write, last_swap = {}, {}
    n = 0
    for kind, text in st.table:
        if kind == "L":
            continue
        n += 1
        m = _re.match(r"v_cvt_pk_\w+ v(\d+),", text)
        if m:
            last_write[int(m.group(1))] = n
        m = _re.match(r"v_permlane16_swap_b32 v(\d+), v(\d+)", text)
        if m:
            for r in (int(m.group(1)), int(m.group(2))):
                assert n - last_write.get(r, -99) > 2, "swap too close behind the pack of v%d" % r
                last_swap[r] = n
        m = _re.match(r"v_mfma_f32_16x16x32_bf16 a\[\d+:\d+\], v\[\d+:\d+\], v\[(\d+):(\d+)\]", text)
        if m:
            for r in range(int(m.group(1)), int(m.group(2)) + 1):
                assert n - last_swap.get(r, -99) > 2, "MFMA too close behind the swap of v%d" % r


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


FAST = False   # set by generate(): the body being emitted is the bounded / single-segment / whole-tile variant


def k_dma(st, L, slot, i, part=3):
    """K loader slot i of this wave -> ring slot `slot` (instruction j = wave + NW i).  part 1 = M0 write only,
    2 = the DMA only (one other instruction must sit between them), 3 = both with an s_nop.  On the ragged last tile
    of a key segment (lane mask S_KRG) the rows past the segment re-fetch its last key (offsets koffL)."""
    if part & 1:
        st.emit("s_add_u32 m0, s%d, %d" % (S_KDST, L.G.KOFF[slot] + 1024 * L.NW * i), "s")
        if not FAST:
            st.emit("v_cndmask_b32_e64 %s, %s, %s, s[%d:%d]" % (vr(L.TX[3]), L.OP["koff%d" % i], L.OP["koffL%d" % i], S_KRG, S_KRG + 1), "v")
        elif part == 3:
            st.emit("s_nop 0", "n")
    if part & 2:   # FAST: the offsets in force (koff, or koffL while the loader sits on a ragged tile) live in KCUR registers
        st.emit("global_load_lds_dwordx4 %s, s[%d:%d]" % (vr(L.KCUR + i) if FAST else vr(L.TX[3]), S_KB, S_KB + 1), "g")


def v_dma(st, L, slot, i, part=3):
    if part & 1:
        st.emit("s_add_u32 m0, s%d, %d" % (S_VDST, (L.G.VOFF[slot] - L.G.VOFF[0]) + 1024 * L.NW * i), "s")
    if part == 3:
        st.emit("s_nop 0", "n")
    if part & 2:
        st.emit("global_load_lds_dwordx4 %s, s[%d:%d]" % (L.OP["voff%d" % i], S_VB, S_VB + 1), "g")


def nslot(L, which):
    return L.NSLOT if which == "k" or not L.PV8 else L.NSLOT_V


def dma_last(st, L, which, slot, uid):
    """last loader slot: only the wave(s) whose instruction index is < NKD / NVD (all of them when NW divides it)"""
    n = nslot(L, which)
    if not (L.LAST_COND if which == "k" or not L.PV8 else L.LAST_COND_V):
        (k_dma if which == "k" else v_dma)(st, L, slot, n - 1)
        return
    lab = ".L@@_%s2%s" % (which, uid)
    st.emit("s_cmp_lt_u32 s%d, %d" % (S_NKW if which == "k" else S_NVW, n), "s")
    st.emit("s_cbranch_scc1 %s" % lab, "s")
    (k_dma if which == "k" else v_dma)(st, L, slot, n - 1)
    st.label(lab)


def advance(st, which, uid, vstep=128, L=None, slot=0):
    """point the loader at its next tile; past the last tile it stays (harmless re-fetch of the last tile)"""
    if FAST:   # branch-free, six SALU instructions, behind the event check (slow path: fast_events())
        sl, sb = (S_KL, S_KB) if which == "k" else (S_VL, S_VB)
        st.emit("s_cmp_eq_u32 s%d, s%d" % (sl, S_KNEXT if which == "k" else S_VNEXT), "s")
        st.emit("s_cbranch_scc1 .L@@_ev%s%s" % (which, uid), "s")
        st.events.append((which, uid, slot, vstep))
        st.label(".L@@_ec%s%s" % (which, uid))
        st.emit("s_add_u32 s%d, s%d, 1" % (S_TMP, sl), "s")
        st.emit("s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NT), "s")
        st.emit("s_cselect_b32 s%d, s%d, s%d" % (sl, S_TMP, sl), "s")
        if which == "k":
            st.emit("s_cselect_b32 s%d, s%d, 0" % (S_TMP, S_KSTEP), "s")
        else:
            st.emit("s_cselect_b32 s%d, %d, 0" % (S_TMP, vstep), "s")
        st.emit("s_add_u32 s%d, s%d, s%d" % (sb, sb, S_TMP), "s")
        st.emit("s_addc_u32 s%d, s%d, 0" % (sb + 1, sb + 1), "s")
        st.label(".L@@_ed%s%s" % (which, uid))
        return
    lab = ".L@@_%sa%s" % (which, uid)
    sl, sb, stt, sj = (S_KL, S_KB, S_KTT, S_KJ) if which == "k" else (S_VL, S_VB, S_VTT, S_VJ)
    st.emit("s_add_u32 s%d, s%d, 1" % (S_TMP, sl), "s")
    st.emit("s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NT), "s")
    st.emit("s_cbranch_scc0 %s" % lab, "s")
    st.emit("s_mov_b32 s%d, s%d" % (sl, S_TMP), "s")
    if which == "k":
        st.emit("s_add_u32 s%d, s%d, s%d" % (sb, sb, S_KSTEP), "s")
    else:
        st.emit("s_add_u32 s%d, s%d, %d" % (sb, sb, vstep), "s")
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
    if FAST:
        return
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
    st.emit("v_cndmask_b32_e64 %s, %s, 0, s[%d:%d]" % (vr(L.TX[2]), vr(L.TX[2]), L.S_ONESMASK, L.S_ONESMASK + 1), "v")
    st.emit("v_cndmask_b32_e64 %s, %s, %s, s[%d:%d]" % (vr(L.TX[2]), vr(L.TX[2]), L.OP["maskval"], S_VRG, S_VRG + 1), "v")
    st.emit("ds_write_b32 %s, %s offset:%d" % (L.OP["onesaddr"], vr(L.TX[2]), L.G.VOFF[slot] - L.G.VOFF[0]), "D")
    st.label(lab)


def dma_group(st, L, which, slot, uid):
    """all LDS-DMA instructions of this wave for one K (or V^T) tile + loader advance (prologue form)"""
    for i in range(nslot(L, which) - 1):
        (k_dma if which == "k" else v_dma)(st, L, slot, i)
    dma_last(st, L, which, slot, uid)
    advance(st, which, uid, L.G.VSTEP, L, slot)
    ragged_masks(st)


def row_write(st, L, slot, mask, uid):
    """FAST event code: the V^T ring slot's ones row (softmax denominator) := the ragged tile's key-validity mask / ones again.
    The ones-row wave only (flag bit 1, set when the launch has a ragged tile); lanes 32..63 rewrite the zero row behind it."""
    lab = ".L@@_rw%s" % uid
    st.emit("s_bitcmp1_b32 s%d, 1" % S_FLG, "s")
    st.emit("s_cbranch_scc0 %s" % lab, "s")
    if mask:
        st.emit("v_mov_b32 %s, %s" % (vr(L.TX[2]), L.OP["maskval"]), "v")
    else:
        st.emit("v_mov_b32 %s, 0x3f803f80" % vr(L.TX[2]), "v")
    st.emit("v_cndmask_b32_e64 %s, %s, 0, s[%d:%d]" % (vr(L.TX[2]), vr(L.TX[2]), L.S_ONESMASK, L.S_ONESMASK + 1), "v")
    st.emit("ds_write_b32 %s, %s offset:%d" % (L.OP["onesaddr"], vr(L.TX[2]), L.G.VOFF[slot] - L.G.VOFF[0]), "D")
    st.label(lab)


def fast_events(st, L):
    """Out-of-line event code of the FAST bodies' loaders, one copy per advance site (the site jumps here when the loader's tile
    index equals its next-event index, and gets a complete advance back).  With E = the index of the current key segment's last
    tile (S_KE / S_VE; segments are tps tiles; a launch is `ragged` when seg_len % 64 != 0, i.e. every E tile is short):
      K loader (two tiles ahead of the MFMAs), at old index E-1: the tile it moves to is E -- its LDS-DMA pieces fetch through the
        clamped per-lane offsets koffL (rows past the segment re-fetch its last key) if ragged; next event at E.  At old index E:
        back to koff, the segment jump, E += tps, next event at E-1.  Past the last tile the loader does not move (and keeps
        its offsets: it re-fetches that tile).
      V^T loader, at old index E: the ring slot that receives tile E in this body gets the validity mask as its ones row (if
        ragged; next event two tiles later: the same slot receives tile E+2 and gets its ones back), then advance + segment jump,
        E += tps.  The wrapper falls back to the general body when segments are shorter than 3 tiles and there are several."""
    for which, uid, slot, vstep in st.events:
        e = lambda t, kind="s": st.emit(t, kind)
        done, common = ".L@@_ed%s%s" % (which, uid), ".L@@_ec%s%s" % (which, uid)
        st.label(".L@@_ev%s%s" % (which, uid))
        if which == "k":
            sl, sb = S_KL, S_KB
            e("s_add_u32 s%d, s%d, 1" % (S_TMP, sl))
            e("s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NT))
            e("s_cbranch_scc0 %s" % done)                                    # clamped at the last tile: nothing moves
            e("s_mov_b32 s%d, s%d" % (sl, S_TMP))
            e("s_add_u32 s%d, s%d, s%d" % (sb, sb, S_KSTEP))
            e("s_addc_u32 s%d, s%d, 0" % (sb + 1, sb + 1))
            e("s_cmp_eq_u32 s%d, s%d" % (S_TMP, S_KE))
            e("s_cbranch_scc0 .L@@_el%s" % uid)
            # entering the segment's last tile
            e("s_mov_b32 s%d, s%d" % (S_KNEXT, S_TMP))
            e("s_cmp_lg_u32 s%d, 0" % S_NRG)
            e("s_cbranch_scc1 %s" % done)
            for i in range(L.NSLOT):
                e("v_mov_b32 %s, %s" % (vr(L.KCUR + i), L.OP["koffL%d" % i]), "v")
            e("s_branch %s" % done)
            st.label(".L@@_el%s" % uid)                                      # leaving it: next segment
            for i in range(L.NSLOT):
                e("v_mov_b32 %s, %s" % (vr(L.KCUR + i), L.OP["koff%d" % i]), "v")
            e("s_add_u32 s%d, s%d, s%d" % (sb, sb, S_KJ))
            e("s_addc_u32 s%d, s%d, s%d" % (sb + 1, sb + 1, S_KJ + 1))
            e("s_add_u32 s%d, s%d, s%d" % (S_KE, S_KE, S_TPS))
            e("s_sub_u32 s%d, s%d, 1" % (S_KNEXT, S_KE))
            e("s_branch %s" % done)
        else:
            sl, sb = S_VL, S_VB
            e("s_cmp_eq_u32 s%d, s%d" % (sl, S_VE))
            e("s_cbranch_scc0 .L@@_er%s" % uid)
            # the loader sits on its segment's last tile, whose LDS-DMA pieces this body has issued into ring slot `slot`
            e("s_add_u32 s%d, s%d, s%d" % (S_VNEXT, sl, S_TPS))
            e("s_cmp_lg_u32 s%d, 0" % S_NRG)
            e("s_cbranch_scc1 .L@@_ej%s" % uid)
            row_write(st, L, slot, True, "m" + uid)
            e("s_add_u32 s%d, s%d, 2" % (S_VNEXT, sl))
            st.label(".L@@_ej%s" % uid)
            e("s_add_u32 s%d, s%d, s%d" % (S_VE, S_VE, S_TPS))
            e("s_add_u32 s%d, s%d, 1" % (S_TMP, sl))
            e("s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NT))
            e("s_cbranch_scc0 %s" % done)
            e("s_mov_b32 s%d, s%d" % (sl, S_TMP))
            e("s_add_u32 s%d, s%d, %d" % (sb, sb, vstep))
            e("s_addc_u32 s%d, s%d, 0" % (sb + 1, sb + 1))
            e("s_add_u32 s%d, s%d, s%d" % (sb, sb, S_VJ))
            e("s_addc_u32 s%d, s%d, s%d" % (sb + 1, sb + 1, S_VJ + 1))
            e("s_branch %s" % done)
            st.label(".L@@_er%s" % uid)                                      # two tiles behind a ragged tile: its slot's ones row back
            row_write(st, L, slot, False, "o" + uid)
            e("s_mov_b32 s%d, s%d" % (S_VNEXT, S_VE))
            e("s_branch %s" % common)


def fast_preamble(st, L):
    """FAST bodies: initial state of the loader events (fast_events())"""
    e = st.emit
    # loader events (fast_events()): the first segment's last tile is tps - 1; a launch that is ONE segment of whole tiles has
    # no event at all; when tile 0 itself is the ragged last tile the K loader starts on the clamped offsets
    e("s_sub_u32 s%d, s%d, 1" % (S_KE, S_TPS))
    e("s_mov_b32 s%d, s%d" % (S_VE, S_KE))
    e("s_mov_b32 s%d, s%d" % (S_VNEXT, S_KE))
    e("s_sub_u32 s%d, s%d, 2" % (S_KNEXT, S_TPS))
    for i in range(L.NSLOT):
        e("v_mov_b32 %s, %s" % (vr(L.KCUR + i), L.OP["koff%d" % i]))
    e("s_cmp_lg_u32 s%d, 0" % S_NRG)
    e("s_cbranch_scc0 .L@@_ini1")
    e("s_cmp_lg_u32 s%d, s%d" % (S_TPS, S_NT))
    e("s_cbranch_scc1 .L@@_ini2")
    e("s_mov_b32 s%d, -1" % S_KNEXT)
    e("s_mov_b32 s%d, -1" % S_VNEXT)
    e("s_branch .L@@_ini2")
    st.label(".L@@_ini1")
    e("s_cmp_lg_u32 s%d, 1" % S_TPS)
    e("s_cbranch_scc1 .L@@_ini2")
    for i in range(L.NSLOT):
        e("v_mov_b32 %s, %s" % (vr(L.KCUR + i), L.OP["koffL%d" % i]))
    st.label(".L@@_ini2")


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
        # Q's padding dim that carries -M: word 0 of the k-step-M_KS fragment, in the half-wave S_HIM selects
        st.emit("v_accvgpr_read_b32 %s, %s" % (vr(t), ar(L.AQ(u, L.G.M_KS))))
        st.emit("s_nop 0", "n")
        st.emit("v_cndmask_b32_e64 %s, %s, %s, s[%d:%d]" % (vr(t), vr(t), vr(pk), S_HIM, S_HIM + 1))
        st.emit("s_nop 0", "n")
        st.emit("v_accvgpr_write_b32 %s, %s" % (ar(L.AQ(u, L.G.M_KS)), vr(t)))
        for t2 in range(2):
            for r in range(16):
                x = vr(L.S(sx, u, t2, r))
                st.emit("v_add_f32 %s, %s, %s" % (x, x, vr(de)))
        if not init and L.PV16:
            # O^T of block u lives in 16-query blocks (lane = query l % 16 of block qb, every lane row): bring alpha of query
            # 16 qb + (l % 16) into all four lane rows -- swap(X, Y) with X = Y = alpha leaves X = alpha of queries 0..15, Y = 16..31
            ax, ay = L.TX[0], L.TX[1]   # (d, n: dead by now)
            st.emit("v_mov_b32 %s, %s" % (vr(ax), vr(al)))
            st.emit("v_mov_b32 %s, %s" % (vr(ay), vr(al)))
            st.emit("s_nop 1", "n")
            st.emit("v_permlane16_swap_b32 %s, %s" % (vr(ax), vr(ay)))
            st.emit("s_nop 1", "n")
            for qb in range(2):
                for db in range(L.G.NDB):
                    for r in range(4):
                        st.emit("v_accvgpr_read_b32 %s, %s" % (vr(T + r), ar(L.AO16(u, qb, db) + r)))
                    st.emit("s_nop 0", "n")
                    for r in range(4):
                        st.emit("v_mul_f32 %s, %s, %s" % (vr(T + r), vr(T + r), vr(ax if qb == 0 else ay)))
                    st.emit("s_nop 0", "n")
                    for r in range(4):
                        st.emit("v_accvgpr_write_b32 %s, %s" % (ar(L.AO16(u, qb, db) + r), vr(T + r)))
        elif not init:
            for dd in range(L.G.NDT):
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
DMAGAP = 1     # LDS-DMA items: one per DMAGAP MFMA shadows (set by --exp dmagapN for the experimental variant)
DMACOST = 12   # issue cycles a shadow's filler budget is charged for an LDS-DMA item


def _check_p_ready(st, L, k):
    """every P.V MFMA of this body's own tile must come AFTER the v_cvt_pk instructions that write its P operand registers
    (the filler windows are hand-set: this is what keeps an edited window honest).  The trailing MFMAs at the top of the body
    consume the PREVIOUS tile's last fragments and are exempt."""
    import re as _re
    start = max(i for i, (kind, text) in enumerate(st.table) if kind == "L" and text.endswith("_entry%d" % k))
    written, swapped = set(), set()
    for kind, text in st.table[start:]:
        if text.startswith("v_cvt_pk"):
            written.add(int(_re.match(r"v_cvt_pk_\w+ v(\d+),", text).group(1)))
        elif text.startswith("v_permlane16_swap"):
            a_, b_ = (int(x) for x in _re.match(r"v_permlane16_swap_b32 v(\d+), v(\d+)", text).groups())
            assert {a_, b_} <= written, "lane-row swap before its registers are packed: " + text
            swapped |= {a_, b_}
        elif text.startswith("v_mfma") and kind in ("M", "F") and ", a[" in text.split(",", 1)[0] + ",":   # accumulating into O^T (AGPRs)
            m = _re.match(r"v_mfma_\w+ a\[\d+:\d+\], v\[\d+:\d+\], v\[(\d+):(\d+)\]", text)
            if m:
                need = set(range(int(m.group(1)), int(m.group(2)) + 1))
                assert need <= written, "P.V MFMA reads P registers %s before their v_cvt_pk: %s" % (sorted(need - written), text)
                assert not L.PV16 or need <= swapped, "P.V MFMA reads P registers before their lane-row swap: " + text


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
    # -- trailing P.V MFMAs of tile t-1 (the last two fragment pairs were read before the barrier); in their shadows the
    #    first LDS-DMA pieces of K(t+2) -> slot cur and V(t+1) -> slot cur^1 (M0 write BEFORE the MFMA: no s_nop)
    pieces = [("k", i) for i in range(nslot(L, "k") - 1)] + [("v", i) for i in range(nslot(L, "v") - 1)]
    pieces = pieces[:(L.NTRAIL + DMAGAP - 1) // DMAGAP]
    done = {"k": sum(1 for w, _ in pieces if w == "k"), "v": sum(1 for w, _ in pieces if w == "v")}

    def piece(pc, part):
        (k_dma if pc[0] == "k" else v_dma)(st, L, cur if pc[0] == "k" else cur ^ 1, pc[1], part)

    for n in range(L.NTRAIL):
        pn = n // DMAGAP if n % DMAGAP == 0 else len(pieces)
        if pn < len(pieces):
            piece(pieces[pn], 1)
        pv_mfma(st, L, (L.G.NPV - L.NTP) * L.MPP + n)
        if pn < len(pieces):
            piece(pieces[pn], 2)
    # -- decision: did some score of tile t exceed the reference by more than 2^THR (VCC from the previous body)?
    if not FAST:   # (fast: the reference is a bound, nothing to decide)
        st.emit("s_cbranch_vccnz .L@@_rare%d" % k)
    st.label(".L@@_entry%d" % k)

    # -- remaining LDS-DMA work, one item per shadow right after the entry point: (emitter, cycles)
    later = []
    for i in range(done["k"], nslot(L, "k") - 1):
        later.append((lambda i=i: k_dma(st, L, cur, i), DMACOST))
    later.append((lambda: dma_last(st, L, "k", cur, uid), DMACOST))
    for i in range(done["v"], nslot(L, "v") - 1):
        later.append((lambda i=i: v_dma(st, L, cur ^ 1, i), DMACOST))
    later.append((lambda: dma_last(st, L, "v", cur ^ 1, uid), DMACOST))
    if not L.PV8 and not FAST:   # pv8: the ones / key-validity row arrives with the V^T tile itself; fast: whole tiles only
        later.append((lambda: ones_row(st, L, cur ^ 1), 8))   # before the V^T loader moves on: S_VRG is tile t+1's
    later.append((lambda: advance(st, "k", uid, 128, L, cur), 20 if FAST else 28))
    later.append((lambda: (advance(st, "v", uid, L.G.VSTEP, L, cur ^ 1), ragged_masks(st)), 20 if FAST else 32))

    # -- fillers paced in CYCLES (v_exp 8, other VALU 4): three classes, each spread uniformly over its window
    cost = lambda kind_: 8.0 if kind_ == "e" else (2.0 if kind_ == "n" else 4.0)
    clsA, clsB, clsC = [], [], []
    for g in range(4):
        for u in range(NU):
            if L.PV16:   # a 32-key half (two key groups) feeds the first P.V pairs: class A = keys 0..31, B = keys 32..63
                (clsA if g < 2 else clsB).extend(exp_group(L, sc, u, g))
            else:
                (clsA if g < 3 else clsB).extend(exp_group(L, sc, u, g))
        if L.PV16 and g % 2 == 1:
            for u in range(NU):
                (clsA if g < 2 else clsB).extend(swap_group(L, u, g // 2))
    ca = [max_chain(L, sn, u, 0, chain_tmp(L, u, 0)) for u in range(NU)]
    cb = [max_chain(L, sn, u, 1, chain_tmp(L, u, 1)) for u in range(NU)]
    clsC.extend([o for grp in zip(*ca) for o in grp])
    n_first_b = len(clsC)
    clsC.extend([o for grp in zip(*cb) for o in grp])
    clsC.extend(lanemax_final(L))
    if FAST:   # no reference max to maintain
        clsC, n_first_b = [], 0
    W = L.FAST_WINDOWS if FAST else L.WINDOWS
    classes = [[clsA, W[0], W[1], 0, 0.0], [clsB, W[2], W[3], 0, 0.0], [clsC, W[4], W[5], 0, 0.0]]
    totals = [sum(cost(o[0]) for o in c[0]) for c in classes]
    mf = [("qk", a) for a in range(L.NQK)] + [("pv", b) for b in range(L.NPVB)]
    for n, (kind, idx) in enumerate(mf):
        i = L.I_QK0 + n
        per = L.MPP if kind == "pv" else NU
        pair, u = idx // per, idx % per
        if kind == "pv" and L.PV8:
            if u == 0:
                st.need(("v", pair))
        elif u == 0 and pair % 2 == 0:   # pairs 2m and 2m+1 with one wait (both were issued >= 3 pairs ago)
            st.need(("k" if kind == "qk" else "v", pair + 1))
        if kind == "qk":
            qk_mfma(st, L, sn, idx)
        else:
            pv_mfma(st, L, idx)
        if safe:
            st.emit("s_nop 7", "n")
        used = 0.0
        # ring reads: after the last MFMA of pair x its ring slot is free -> read pair x + 4
        if u == per - 1:
            if kind == "qk":
                NR = L.G.NPK_READ
                if pair + 4 < NR:
                    k_read(st, L, cur ^ 1, pair + 4, ("k", pair + 4))
                    used += 4
                elif pair < NR and pair + 4 - NR < L.RD:
                    v_read(st, L, cur, pair + 4 - NR, ("v", pair + 4 - NR))   # first V pairs behind the last K pairs
                    used += 8 if L.PV8 else 4
            elif pair + L.RD < L.G.NPV:
                v_read(st, L, cur, pair + L.RD, ("v", pair + L.RD))
                used += 8 if L.PV8 else 4
        if later and n % DMAGAP == 0:
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
                if used >= (N2_BUDGET["pv8"] if kind == "pv" and L.PV8 else (N2_BUDGET["pv16"] if kind == "pv" and L.PV16 else N2_BUDGET["m32"])) and i < last:
                    break
                st.emit(text, kind_)
                used += cost(kind_)
                c[4] += cost(kind_)
                c[3] += 1
    for c in classes:
        assert c[3] == len(c[0]), "unscheduled filler work"
    assert not later
    _check_p_ready(st, L, k)
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


def generate(L, safe=False, ablate=frozenset(), fast=False):
    assert not L.G.NOM or fast, "the padding k-step may only be dropped in the bounded (FAST) body"
    """fast: the bounded / single-segment / whole-tile variant (see the header); not for the fp8 P.V kernels, whose e4m3 P
    needs the reference max within 2^8 of the true one"""
    global FAST
    assert not (fast and L.PV8)
    FAST = fast
    try:
        return _generate(L, safe, ablate)
    finally:
        FAST = False


def _generate(L, safe, ablate):
    st = Stream(ablate)
    e = st.emit
    # ---- copy the mutable scalars into asm-owned SGPRs
    e("s_mov_b64 s[%d:%d], %s" % (S_KB, S_KB + 1, L.OP["kbase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_VB, S_VB + 1, L.OP["vbase"]))
    e("s_mov_b32 s%d, %s" % (S_KSTEP, L.OP["kstep"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_KJ, S_KJ + 1, L.OP["kjump"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_VJ, S_VJ + 1, L.OP["vjump"]))
    for sreg, name in ((S_TPS, "tps"), (S_NT, "nt"), (S_KDST, "kdst"), (S_VDST, "vdst"), (S_NKW, "nkw"), (S_NVW, "nvw")):
        if name in L.OP:
            e("s_mov_b32 s%d, %s" % (sreg, L.OP[name]))
    if L.PV16:   # tps | nt << 16 and nvw | nkw << 16 arrive packed (operand slots)
        e("s_and_b32 s%d, %s, 0xffff" % (S_TPS, L.OP["tpsnt"]))
        e("s_lshr_b32 s%d, %s, 16" % (S_NT, L.OP["tpsnt"]))
        e("s_lshr_b32 s%d, %s, 16" % (S_NKW, L.OP["nkvw"]))
        e("s_and_b32 s%d, %s, 0xffff" % (S_NVW, L.OP["nkvw"]))
    for sreg in (S_T, S_KTT, S_VTT, S_KL, S_VL):
        e("s_mov_b32 s%d, 0" % sreg)
    if L.G.M_HI:       # S_HIM = the half-wave whose lanes hold Q's M-carrying padding dim
        e("s_mov_b32 s%d, 0" % S_HIM)
        e("s_mov_b32 s%d, -1" % (S_HIM + 1))
    else:
        e("s_mov_b32 s%d, -1" % S_HIM)
        e("s_mov_b32 s%d, 0" % (S_HIM + 1))
        e("s_mov_b32 s%d, 0" % S_HI2)
        e("s_mov_b32 s%d, -1" % (S_HI2 + 1))
        e("v_mov_b32 %s, 0x3f80" % vr(L.KC0))          # K fragment of the padding k-step: dim HD = 1.0, the rest 0
        for i in range(1, 4):
            e("v_mov_b32 %s, 0" % vr(L.KC0 + i))
    # nvw = valid V^T loader slots | (no ragged tile in this launch) << 8 | (this wave maintains the ones rows) << 9
    e("s_lshr_b32 s%d, s%d, 8" % (S_FLG, S_NVW))
    e("s_and_b32 s%d, s%d, 1" % (S_NRG, S_FLG))
    e("s_and_b32 s%d, s%d, 0xff" % (S_NVW, S_NVW))
    if not FAST:
        ragged_mask(st, "k")
        ragged_mask(st, "v")
    else:
        fast_preamble(st, L)
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
    for p in range(L.G.NPK):
        if p < L.G.NPK_READ:
            k_read(st, L, 0, p, ("k", p))
            st.need(("k", p))
        for u in range(L.NU):
            qk_mfma(st, L, L.SA0, p * L.NU + u)
    e("s_barrier")                      # every wave has read K0: slot 0 may be refilled
    e("s_nop 15")
    e("s_nop 15")
    if not FAST:   # (fast: Q's padding dim already carries the bound, the scores above are final)
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
    for k in range(0 if FAST else 2):
        st.label(".L@@_rare%d" % k)
        fixup(st, L, L.SA0 if k == 0 else L.SB0, init=False)
        e("s_branch .L@@_entry%d" % k)
    if FAST:
        e("s_branch .L@@_exit")   # (never reached: both bodies end in branches)
        fast_events(st, L)
    # ---- exit: the trailing P.V MFMAs of the last tile
    st.label(".L@@_exit")
    for n in range(L.NTRAIL):
        pv_mfma(st, L, (L.G.NPV - L.NTP) * L.MPP + n)
    e("s_nop 15")
    e("s_nop 15")
    e("v_mov_b32 %s, %s" % (L.OP["m0out"], vr(L.MM[0])))
    e("v_mov_b32 %s, %s" % (L.OP["m1out"], vr(L.MM[1])))
    return st


# ================================================================================================ WIDE layout (round 4)
class LayoutW(Layout):
    """head_dim 72, FAST (bounded) body only: 4 waves x 128 query rows = 512-row workgroups, 32-key SUB-tiles.

    A wave owns FOUR 32-row query blocks u'' = 2 ph + u (ph = "phase", u in {0, 1}) and a loop body handles ONE 32-key half h of
    a 64-key tile.  Per body and wave that is the same 20 + 40 MFMAs, the same 64 exp2 / 32 packs / 16 lane-row swaps and the
    same register budget as the 2-block layout's body over a whole tile (what that layout indexes by key half t2 is the phase
    here: score set S(set, u, ph), packed P of block u'' in PB(u, 2 ph + qb)) -- but every K fragment now feeds 4 MFMAs instead
    of 2 and every V^T fragment 8 instead of 4 (five resident V^T slots: phase 1 re-uses the fragments phase 0 read), and a
    64-key tile's LDS-DMA pieces and loader advances are spread over two bodies: per 1280 matrix cycles 10 fragment reads instead
    of 20, ~2.4 LDS-DMA pieces instead of ~4.8, one loader advance instead of two.  O^T: 160 AGPRs (4 blocks), Q: 80.
    Ring protocol (2 K slots, 2 V^T slots, one barrier per body): body (t, h) = P.V of half h of tile t (V^T slot t % 2) beside
    QK^T of the NEXT half -- (t, 1) from K slot t % 2 when h == 0, (t + 1, 0) from the other slot when h == 1.  K(t + 2) streams
    into K(t)'s slot during (t, 1) (K(t) was last read in (t, 0)); V^T(t + 1) into V^T(t - 1)'s slot during (t, 0)."""

    def __init__(self):
        Layout.__init__(self, 2, 72, pv16=True)
        G = self.G
        self.WIDE = True
        self.NUW = 4                                   # query blocks per wave
        self.A_O0 = 0
        self.A_Q0 = 4 * G.NDB * 2 * self.NUW           # 160
        self.A_END = self.A_Q0 + 4 * G.NKS * self.NUW  # 240
        self.VR0 = self.KR0 + 16                       # five V^T fragment slots (20 registers)
        self.TMP0 = self.VR0 + 20
        self.KCUR = self.TMP0
        self.MM = [self.TMP0 + 4, self.TMP0 + 5]
        self.MT = [self.TMP0 + 6, self.TMP0 + 7]
        self.TX = [self.TMP0 + 8 + i for i in range(4)]
        self.V_END = self.TMP0 + 12
        assert self.V_END <= 256 and self.A_END <= 256

    def AO16W(self, u2, qb, db):
        return self.A_O0 + ((u2 * 2 + qb) * self.G.NDB + db) * 4

    def AQW(self, u2, ks):
        return self.A_Q0 + (u2 * self.G.NKS + ks) * 4


def kw_read(st, L, slot, half, ks, tag):
    """K fragment (32 keys of `half`, k-step ks) of the tile in ring slot `slot` -> K ring slot ks % 4"""
    dst = L.KR0 + (ks % 4) * 4
    if ks < 4:
        st.ds_read(dst, L.OP["fo%d" % ks], L.G.KOFF[slot] + half * 4096, tag)
    else:
        st.ds_read(dst, L.OP["kc%d" % half], L.G.KOFF[slot], tag)


def qkw_mfma(st, L, sn, a):
    p, u = a // 2, a % 2
    ks, ph = p // 2, p % 2
    dst = vr(L.S(sn, u, ph), 16)
    st.emit("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (dst, vr(L.KR0 + (ks % 4) * 4, 4), ar(L.AQW(2 * ph + u, ks), 4),
                                                       "0" if ks == 0 else dst), "M")


def pvw_mfma(st, L, b):
    r, sub = b // 4, b % 4
    u, qb = sub // 2, sub % 2
    ph, db = r // L.G.NDB, r % L.G.NDB
    dst = ar(L.AO16W(2 * ph + u, qb, db), 4)
    st.emit("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (dst, vr(L.VR0 + db * 4, 4), vr(L.PB(u, 2 * ph + qb), 4), dst), "X")


WIDE_EXP = {"gap": 1, "win": None, "vm": 0, "swapnop": False, "kwait2": True, "dmafill": True, "budget": (9, 24), "pretrail": 0}   # production values; --wide-exp flips them   # --wide-exp (experiments): DMA piece spacing in the trailing shadows, filler windows, swap placement


def body_wide(st, L, k, h, safe=False):
    """body (k, h): P.V of key half h of the tile in ring slot k (its scores: set A when h == 0, set B when h == 1) beside QK^T
    of the next half into the other set"""
    G = L.G
    sc, sn = (L.SA0, L.SB0) if h == 0 else (L.SB0, L.SA0)
    nslot_, nhalf = (k, 1) if h == 0 else (k ^ 1, 0)
    uid = "w%d%d" % (k, h)
    which = "v" if h == 0 else "k"                 # the loader this body drives: V^T(t + 1) -> slot k ^ 1 / K(t + 2) -> slot k
    dslot = (k ^ 1) if h == 0 else k
    st.label(".L@@_body%d%d" % (k, h))
    for ks in range(4):
        kw_read(st, L, nslot_, nhalf, ks, ("k", ks))
    dma = k_dma if which == "k" else v_dma
    gap = WIDE_EXP["gap"]
    pieces = list(range(nslot(L, which) - 1))[:(L.NTRAIL + gap - 1) // gap]
    cost = lambda kind_: 8.0 if kind_ == "e" else (2.0 if kind_ == "n" else 4.0)
    clsA, clsB = [], []
    for g in range(4):
        for u in range(2):
            (clsA if g < 2 else clsB).extend(exp_group(L, sc, u, g))
        if g % 2 == 1:
            for u in range(2):
                (clsA if g < 2 else clsB).extend(swap_group(L, u, g // 2, guard=WIDE_EXP["swapnop"]))
    # round 5: the trailing P.V shadows of the PREVIOUS half (8 x 16 cycles that carried only LDS-DMA M0 writes) take the first
    # exponentials of THIS half's phase-0 scores -- complete since the previous body's QK^T, touched by nobody else: exp2 in place --
    # one per shadow, out of the QK^T shadows that were the body's fullest.  A body entered at its entry label (the prologue's jump into
    # body (0, 0)) has not run them: `pre` is returned and the prologue emits them itself.
    pre = []
    for n in range(L.NTRAIL):
        pn = n // gap if n % gap == 0 else len(pieces)
        if pn < len(pieces):
            dma(st, L, dslot, pieces[pn], 1)
        pvw_mfma(st, L, 32 + n)                    # pairs 8, 9 = (phase 1, row blocks 3, 4) of the previous body
        if pn < len(pieces):
            dma(st, L, dslot, pieces[pn], 2)
        for _ in range(WIDE_EXP["pretrail"]):
            if len(pre) < len(clsA) and clsA[len(pre)][0] == "e":
                st.emit(clsA[len(pre)][1], "e")
                pre.append(clsA[len(pre)])
    st.label(".L@@_entry%d%d" % (k, h))
    later = []
    deferred = []      # the LDS-DMA instruction of a piece whose M0 write was just emitted: goes behind the next filler instruction
    for i in range(len(pieces), nslot(L, which) - 1):
        if WIDE_EXP["dmafill"]:
            later.append((lambda i=i: (dma(st, L, dslot, i, 1), deferred.append(lambda i=i: dma(st, L, dslot, i, 2))), DMACOST))
        else:
            later.append((lambda i=i: dma(st, L, dslot, i), DMACOST))
    later.append((lambda: dma_last(st, L, which, dslot, uid), DMACOST))
    later.append((lambda: advance(st, which, uid, G.VSTEP, L, dslot), 20))
    W = WIDE_EXP["win"] or L.FAST_WINDOWS
    classes = [[clsA, W[0], W[1], len(pre), sum(cost(o[0]) for o in pre)], [clsB, W[2], W[3], 0, 0.0]]
    totals = [sum(cost(o[0]) for o in c[0]) for c in classes]
    mf = [("qk", a) for a in range(20)] + [("pv", b) for b in range(32)]
    for n, (kind, idx) in enumerate(mf):
        i = L.I_QK0 + n
        used = 0.0
        if kind == "qk":
            p, u = idx // 2, idx % 2
            ks, ph = p // 2, p % 2
            if u == 0 and ph == 0:
                if not WIDE_EXP["kwait2"]:
                    st.need(("k", ks))
                elif WIDE_EXP["kwait2"] == 4:
                    if ks in (0, 4):       # experiment: one wait for the four fragments read at the top, one for the fifth
                        st.need(("k", 3 if ks == 0 else 4))
                elif ks % 2 == 0:          # one wait per two k-steps (the fragments of both were issued long before)
                    st.need(("k", min(ks + 1, G.NKS - 1)))
            qkw_mfma(st, L, sn, idx)
            if u == 1 and ph == 1:                 # both phases of this k-step issued: its K ring slot is free
                if ks + 4 < G.NKS:
                    kw_read(st, L, nslot_, nhalf, ks + 4, ("k", ks + 4))
                    used += 4
                if ks >= 1:                        # V^T fragments of row blocks 0..3 behind k-steps 1..4 (slots 3, 4: free since the trailing MFMAs)
                    st.ds_read(L.VR0 + (ks - 1) * 4, L.OP["vo%d" % h], G.VOFF[k] + (ks - 1) * 2048, ("v", ks - 1))
                    used += 4
        else:
            r, sub = idx // 4, idx % 4
            if sub == 0 and r < G.NDB and r % 2 == 0:
                st.need(("v", min(r + 1, G.NDB - 1)))
            pvw_mfma(st, L, idx)
            if r == 0 and sub == 3:                # the fifth V^T fragment (row block 4: dims 64..71, ones row)
                st.ds_read(L.VR0 + 4 * 4, L.OP["vo%d" % h], G.VOFF[k] + 4 * 2048, ("v", 4))
                used += 4
        if safe:
            st.emit("s_nop 7", "n")
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
                if totals[ci] * frac - c[4] <= 0 and i < last:
                    break
                if used >= (WIDE_EXP["budget"][0] if kind == "pv" else WIDE_EXP["budget"][1]) and i < last:   # filler cycles per 16 / 32-cycle shadow (round 5: 9 / 24 -- an MFMA's own issue takes 4.4 cycles and v_exp_f32 8.5, not 8: with round 4's 13 / 30 the shadows were over-subscribed; in-step A/B -0.8 % per launch on two boxes, profiles/r05n_attn_filler_budget_ab.jsonl)
                    break
                st.emit(text, kind_)
                used += cost(kind_)
                c[4] += cost(kind_)
                c[3] += 1
                if deferred:               # one instruction now sits between the M0 write and its LDS-DMA
                    deferred.pop(0)()
        if deferred:                       # no filler in this shadow: the wait state the hardware needs
            st.emit("s_nop 0", "n")
            deferred.pop(0)()
    for c in classes:
        assert c[3] == len(c[0]), "unscheduled filler work"
    assert not later and not deferred
    _check_p_ready_wide(st, k, h)
    # the pieces this body issued are not needed before the body after next (K(t + 2) from (t, 1): read from (t + 1, 1) on; V^T(t + 1)
    # from (t, 0): read from (t + 1, 0) on): the wait in front of the barrier only has to retire the PREVIOUS body's pieces
    st.emit("s_waitcnt vmcnt(%d) lgkmcnt(0)" % WIDE_EXP["vm"], "w")
    st.pending = []
    st.emit("s_barrier", "B")
    if h == 1:
        st.emit("s_add_u32 s%d, s%d, 1" % (S_T, S_T), "s")
        st.emit("s_cmp_lt_u32 s%d, s%d" % (S_T, S_NT), "s")
        st.emit("s_cbranch_scc0 .L@@_exit", "s")
        if k == 1:
            st.emit("s_branch .L@@_body00", "s")
    return which, dslot, pieces, pre


def _check_p_ready_wide(st, k, h):
    """body_wide's filler windows are hand-set: every P.V MFMA of the body's own half comes after the packs and the lane-row swap of
    its P operand; every MFMA after the s_waitcnt that covers its LDS operand is checked by Stream.need's bookkeeping"""
    import re as _re
    start = max(i for i, (kind, text) in enumerate(st.table) if kind == "L" and text.endswith("_entry%d%d" % (k, h)))
    written, swapped = set(), set()
    for kind, text in st.table[start:]:
        if text.startswith("v_cvt_pk"):
            written.add(int(_re.match(r"v_cvt_pk_\w+ v(\d+),", text).group(1)))
        elif text.startswith("v_permlane16_swap"):
            a_, b_ = (int(x) for x in _re.match(r"v_permlane16_swap_b32 v(\d+), v(\d+)", text).groups())
            assert {a_, b_} <= written, "lane-row swap before its registers are packed: " + text
            swapped |= {a_, b_}
        elif text.startswith("v_mfma_f32_16x16x32"):
            m = _re.match(r"v_mfma_\w+ a\[\d+:\d+\], v\[\d+:\d+\], v\[(\d+):(\d+)\]", text)
            need = set(range(int(m.group(1)), int(m.group(2)) + 1))
            assert need <= written and need <= swapped, "P.V MFMA before its P operand is ready: " + text


def generate_wide(L, safe=False):
    global FAST
    FAST = True
    try:
        return _generate_wide(L, safe)
    finally:
        FAST = False


def _generate_wide(L, safe):
    st = Stream()
    e = st.emit
    G = L.G
    e("s_mov_b64 s[%d:%d], %s" % (S_KB, S_KB + 1, L.OP["kbase"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_VB, S_VB + 1, L.OP["vbase"]))
    e("s_mov_b32 s%d, %s" % (S_KSTEP, L.OP["kstep"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_KJ, S_KJ + 1, L.OP["kjump"]))
    e("s_mov_b64 s[%d:%d], %s" % (S_VJ, S_VJ + 1, L.OP["vjump"]))
    e("s_mov_b32 s%d, %s" % (S_KDST, L.OP["kdst"]))
    e("s_mov_b32 s%d, %s" % (S_VDST, L.OP["vdst"]))
    e("s_and_b32 s%d, %s, 0xffff" % (S_TPS, L.OP["tpsnt"]))
    e("s_lshr_b32 s%d, %s, 16" % (S_NT, L.OP["tpsnt"]))
    e("s_lshr_b32 s%d, %s, 16" % (S_NKW, L.OP["nkvw"]))
    e("s_and_b32 s%d, %s, 0xffff" % (S_NVW, L.OP["nkvw"]))
    for sreg in (S_T, S_KL, S_VL):
        e("s_mov_b32 s%d, 0" % sreg)
    e("s_mov_b32 s%d, 0" % S_HIM)
    e("s_mov_b32 s%d, -1" % (S_HIM + 1))
    e("s_lshr_b32 s%d, s%d, 8" % (S_FLG, S_NVW))
    e("s_and_b32 s%d, s%d, 1" % (S_NRG, S_FLG))
    e("s_and_b32 s%d, s%d, 0xff" % (S_NVW, S_NVW))
    fast_preamble(st, L)
    for u in range(2):
        e("v_mov_b32 %s, 0" % vr(L.MM[u]))
    for r in range(L.A_O0, L.A_Q0):
        e("v_accvgpr_write_b32 %s, 0" % ar(r))
    dma_group(st, L, "k", 0, "p0")
    dma_group(st, L, "v", 0, "p1")
    dma_group(st, L, "k", 1, "p2")
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    for ks in range(G.NKS):                     # scores of (tile 0, half 0) -> set A
        kw_read(st, L, 0, 0, ks, ("k", ks))
        st.need(("k", ks))
        for ph in range(2):
            for u in range(2):
                qkw_mfma(st, L, L.SA0, (ks * 2 + ph) * 2 + u)
    e("s_nop 15")
    e("s_nop 15")
    which, dslot, pieces, pre = body_wide(Stream(), L, 0, 0)
    for i in pieces:                            # what body (0, 0) does before its entry point
        (k_dma if which == "k" else v_dma)(st, L, dslot, i)
    for kind_, text in pre:                     # (incl. the exponentials its trailing shadows carry: the two s_nop 15 above cover the QK^T latency)
        e(text)
    for ks in range(4):
        kw_read(st, L, 0, 1, ks, ("k", ks))
    e("s_branch .L@@_entry00")
    st.in_body = True
    for k in range(2):
        for h in range(2):
            st.pending = []
            body_wide(st, L, k, h, safe)
    st.in_body = False
    fast_events(st, L)
    if not WIDE_EXP["swapnop"]:
        _check_swap_distance(st)
    st.label(".L@@_exit")
    for n in range(L.NTRAIL):
        pvw_mfma(st, L, 32 + n)
    e("s_nop 15")
    e("s_nop 15")
    e("v_mov_b32 %s, %s" % (L.OP["m0out"], vr(L.MM[0])))
    e("v_mov_b32 %s, %s" % (L.OP["m1out"], vr(L.MM[1])))
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table", type=int, default=0, help="print the schedule of layout NU (1 or 2) shadow by shadow")
    ap.add_argument("--hd", type=int, default=72, help="head dim of the schedule --table prints")
    ap.add_argument("--pv8", action="store_true", help="--table: the fp8 P.V variant")
    ap.add_argument("--exp", default="", help="emit an EXPERIMENTAL body in place of the production one (never into the shipped tree: "
                    "point --out at a scratch copy of csrc, see tools/make_ablated_libs.sh): safe = hazard-padded debug schedule | "
                    "dmagapN[cC] = LDS-DMA items every N shadows | timing ablations joined by + (noexp nobar nodma nolds novalu "
                    "nomfma norare nocvt nomax pv80: WRONG RESULTS)")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "open_sora_amd", "csrc"))
    ap.add_argument("--no-pv16", action="store_true", help="experiment: head_dim 72 with the P.V product on 32x32x16 MFMAs (96 padded rows) as in "
                    "rounds 1-2; changes the V^T key order the kernel expects, so only for timing runs with matching wrappers")
    ap.add_argument("--fast-windows", default="", help="experiment: hd:a0,a1,b0,b1 filler windows of the FAST body (layout NU = 2), e.g. 72:1,33,30,42")
    ap.add_argument("--fast-exp", default="", help="experiment: like --exp, applied to the FAST bodies only")
    ap.add_argument("--n2-budget", default="", help="experiment: m32,pv16,pv8 filler-cycle budgets of the 256-row bodies' MFMA shadows (production 24,9,60)")
    ap.add_argument("--wide-exp", default="", help="experiment on the WIDE head_dim-72 body: gapN (LDS-DMA pieces every N trailing shadows) and / or "
                    "win:a0,a1,b0,b1 (filler windows), joined by +")
    args = ap.parse_args()
    if args.n2_budget:
        N2_BUDGET.update(zip(("m32", "pv16", "pv8"), (int(x) for x in args.n2_budget.split(","))))
    for tok in [t for t in args.wide_exp.split("+") if t]:
        if tok.startswith("gap"):
            WIDE_EXP["gap"] = int(tok[3:])
        elif tok == "kwait4":
            WIDE_EXP["kwait2"] = 4
        elif tok in ("swapnop", "kwait1", "nodmafill"):
            WIDE_EXP[{"swapnop": "swapnop", "kwait1": "kwait2", "nodmafill": "dmafill"}[tok]] = tok == "swapnop"
        elif tok.startswith("vm"):
            WIDE_EXP["vm"] = int(tok[2:])
        elif tok.startswith("pretrail"):    # exponentials of the body's own phase-0 scores per trailing P.V shadow (0: none)
            WIDE_EXP["pretrail"] = int(tok[8:])
        elif tok.startswith("budget:"):     # filler-cycle budget of a P.V / QK^T shadow (round 5: the MFMA's own issue takes 4.4 cycles)
            WIDE_EXP["budget"] = tuple(int(x) for x in tok[7:].split(","))
        elif tok.startswith("win:"):
            w_ = [int(x) for x in tok[4:].split(",")]
            WIDE_EXP["win"] = w_ + [w_[3], w_[3]]
    if args.fast_windows:
        for spec_ in args.fast_windows.split(";"):
            hd_, w_ = spec_.split(":")
            w_ = [int(x) for x in w_.split(",")]
            FAST_WINDOWS_OVERRIDE[(int(hd_), 2)] = w_ + [w_[3], w_[3]]
    # shipped: the 4 waves x 64 rows layout (NU = 2) of both head dims, bf16 and fp8 P.V.  The 8 waves x 32 rows layout
    # (NU = 1, head_dim 72) tied it in rounds 1-2 and stays a generator option (--table 1) without a shipped body.
    layouts = [(72, 2, False), (128, 2, False), (72, 2, True), (128, 2, True)]
    mk = lambda nu, hd, pv8: Layout(nu, hd, pv8, pv16=(hd == 72 and not pv8 and not args.no_pv16))
    tagof = lambda hd, pv8: "%d%s" % (hd, "p8" if pv8 else "")
    global DMAGAP, DMACOST
    safe = args.exp == "safe"
    gap = args.exp.startswith("dmagap")
    ablate = frozenset() if (safe or gap or not args.exp) else frozenset(args.exp.split("+"))
    if (args.exp or args.fast_windows or args.fast_exp or args.no_pv16 or args.wide_exp or args.n2_budget) and os.path.realpath(args.out) == os.path.realpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "open_sora_amd", "csrc")):
        raise SystemExit("--exp bodies are experiments: give --out a scratch directory, not the shipped csrc")
    if args.table:
        L = mk(args.table, args.hd, args.pv8) if args.table == 2 else Layout(args.table, args.hd, args.pv8)
        st = generate(L, False, frozenset())
        row = []
        for kind, text in st.table:
            if kind in ("M", "F", "X"):
                print("".join(row)); row = [kind + " "]
            elif kind == "L":
                print("".join(row)); row = []; print(text + ":")
            else:
                row.append({"x": "v"}.get(kind, kind))
        print("".join(row))
        return
    for hd, nu, pv8 in layouts:
        L = mk(nu, hd, pv8)
        DMAGAP, DMACOST = 1, 12
        if gap:
            spec = args.exp[len("dmagap"):].split("c")
            DMAGAP, DMACOST = int(spec[0]), int(spec[1]) if len(spec) > 1 else 12
        st = generate(L, safe, ablate)
        DMAGAP, DMACOST = 1, 12
        what = "production" if not args.exp else ("hazard-padded (debug)" if safe else ("schedule experiment " + args.exp if gap else
                                                                                         "timing ablation " + "+".join(sorted(ablate))))
        with open(os.path.join(args.out, "attention_asm%s_n%d_v0.inc" % (tagof(hd, pv8), nu)), "w") as f:
            f.write("// GENERATED by tools/gen_attn_asm.py -- do not edit.  head_dim %d%s, layout NU=%d: %s\n" %
                    (hd, ", fp8 P.V" if pv8 else "", nu, what))
            for ln in st.lines:
                f.write('"%s\\n"\n' % ln.replace("@@", "osk%sn%dv0_%%=" % (tagof(hd, pv8), nu)))
        if not pv8:   # the bounded / single-segment / whole-tile body of the same layout
            fexp = args.fast_exp or args.exp
            if fexp.startswith("dmagap"):
                spec = fexp[len("dmagap"):].split("c")
                DMAGAP, DMACOST = int(spec[0]), int(spec[1]) if len(spec) > 1 else 12
            Lf = Layout(nu, hd, pv8, pv16=L.PV16, nom=(hd == 128))   # hd 128: the padding k-step is dropped too
            stf = generate(Lf, safe, ablate, fast=True)
            DMAGAP, DMACOST = 1, 12
            with open(os.path.join(args.out, "attention_asm%s_n%d_f0.inc" % (tagof(hd, pv8), nu)), "w") as f:
                f.write("// GENERATED by tools/gen_attn_asm.py -- do not edit.  head_dim %d, layout NU=%d, FAST body (score bound, one "
                        "segment of whole tiles): %s\n" % (hd, nu, what))
                for ln in stf.lines:
                    f.write('"%s\\n"\n' % ln.replace("@@", "osk%sn%df0_%%=" % (tagof(hd, pv8), nu)))
    # the wide layout of head_dim 72 (4 query blocks per wave, 32-key sub-tiles): FAST body only
    LW = LayoutW()
    stw = generate_wide(LW, safe)
    with open(os.path.join(args.out, "attention_asm72w_f0.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_attn_asm.py -- do not edit.  head_dim 72, WIDE layout (4 waves x 128 query rows, one 32-key half per "
                "body), FAST body: %s\n" % ("production" if not args.exp else "experiment " + args.exp))
        for ln in stw.lines:
            f.write('"%s\\n"\n' % ln.replace("@@", "osk72wf0_%="))
    # register / operand contract for the wrapper
    with open(os.path.join(args.out, "attention_asm_regs.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_attn_asm.py -- do not edit.\n")
        clobw = ['"v%d"' % i for i in range(LW.V_FIRST, LW.V_END)] + ['"a%d"' % i for i in range(0, LW.A_END)] + \
                ['"s%d"' % i for i in range(S_FIRST, S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']
        f.write("#define OSK72W_CLOBBERS %s\n" % ", ".join(clobw))
        f.write("#define OSK72W_A_CLOBBERS %s\n" % ", ".join('"a%d"' % i for i in range(0, LW.A_END)))
        f.write("#define OSK72W_NSLOT %d\n" % LW.NSLOT)
        # register map the wrapper binds as asm operands (acc_quads.h): Q fragment words of block u = AGPRs AQ(u) .. + 4 NKS,
        # O^T of (u, 16-query block qb) = AGPRs 4 NDB (2 u + qb) .. (row block db, register i at + 4 db + i)
        for u in range(LW.NUW):
            f.write("#define OSK72W_AQ%d %d\n" % (u, LW.AQW(u, 0)))
            for qb in range(2):
                assert all(LW.AO16W(u, qb, db) == 4 * LW.G.NDB * (2 * u + qb) + 4 * db for db in range(LW.G.NDB))
        f.write("#define OSK72W_AO_REGS %d\n" % (4 * LW.G.NDB * 2 * LW.NUW))
        for hd, pv8 in sorted({(h, p8) for h, _, p8 in layouts}):
            G = mk(2, hd, pv8).G
            P = "OSK%s_" % tagof(hd, pv8).upper()
            f.write("#define %sSMEM %d\n#define %sCONST_OFF %d\n" % (P, G.SMEM, P, G.CONST_OFF))
            f.write("#define %sKTILE %d\n#define %sVTILE %d\n#define %sVOFF0 %d\n#define %sKOFF0 %d\n" % (P, G.KTILE, P, G.VTILE, P, G.VOFF[0], P, G.KOFF[0]))
            f.write("#define %sNKS %d\n#define %sNDT %d\n#define %sNKD %d\n#define %sKIMG %d\n" % (P, G.NKS, P, G.NDT, P, G.NKD, P, G.KIMG))
            if pv8:
                f.write("#define %sRP %d\n#define %sNVD %d\n" % (P, G.RP, P, G.NVD))
        for hd, nu, pv8 in layouts:
            L = mk(nu, hd, pv8)
            G = L.G
            P = "OSK%sN%d_" % (tagof(hd, pv8).upper(), nu)
            if L.PV16:
                f.write("#define %sPV16 1\n#define OSK%s_NDB %d\n" % (P, tagof(hd, pv8).upper(), G.NDB))
            clob = ['"v%d"' % i for i in range(L.V_FIRST, L.V_END)] + ['"a%d"' % i for i in range(0, L.A_END)] + \
                   ['"s%d"' % i for i in range(S_FIRST, S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']
            f.write("#define %sCLOBBERS %s\n" % (P, ", ".join(clob)))
            f.write("#define %sA_CLOBBERS %s\n" % (P, ", ".join('"a%d"' % i for i in range(0, L.A_END))))
            f.write("#define %sNSLOT %d\n" % (P, L.NSLOT))
            if pv8:
                f.write("#define %sNSLOT_V %d\n" % (P, L.NSLOT_V))
            # register map the wrapper binds as asm operands (acc_quads.h): Q fragment words of block u = AGPRs AQ(u) .. + 4 NKS
            for u in range(nu):
                f.write("#define %sAQ%d %d\n" % (P, u, L.AQ(u, 0)))
            if L.PV16:   # O^T of (u, 16-query block qb) = AGPRs 4 NDB (2 u + qb) .. (row block db, register i at + 4 db + i)
                assert all(L.AO16(u, qb, db) == 4 * G.NDB * (2 * u + qb) + 4 * db for u in range(nu) for qb in range(2) for db in range(G.NDB))
                f.write("#define %sAO_REGS %d\n" % (P, 4 * G.NDB * 2 * nu))
                continue
            assert all(L.AO(u, d) == 16 * (u * G.NDT + d) for u in range(nu) for d in range(G.NDT))   # O^T row tile (u, d): 16 registers
            f.write("#define %sAO_REGS %d\n" % (P, 16 * G.NDT * nu))


if __name__ == "__main__":
    main()
