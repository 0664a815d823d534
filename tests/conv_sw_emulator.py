"""TEST INFRASTRUCTURE -- CPU models of the sliding-window CausalConv3d K loop (tools/gen_conv_sw_asm.py, csrc/conv3d_256.hip::
convsw_kernel).  Two independent executions of the GENERATED instruction stream:

  * `check_schedule`: one representative wave, symbolic.  Scalar instructions run on concrete pointer values (so the address
    arithmetic -- tap advance, channel-block wrap, the clamps of the last iteration -- is executed, not assumed), LDS regions
    carry (channel block, tap) / (channel block, frame) symbols, and the in-order lgkmcnt / vmcnt queues and barrier epochs are
    tracked: a fragment used before its wait, a region read before its fill is published by a barrier, a region refilled while a
    wave may still read it, a stale M0, or an accumulator tile that misses / repeats a (block, tap) product is an error.
  * `emulate_tile`: all four waves in lock step, numeric, lane by lane: LDS as a byte array, LDS-DMA pieces gather 16 bytes per
    lane from the tensors, ds_read_b128 / v_mfma_f32_16x16x32_bf16 with the hardware's lane maps, with the lane formulas of the
    C++ wrapper re-stated in `wrapper_operands` (keep the two in sync: the formulas are the contract).  Timing is ignored here
    (a piece lands when it is issued) -- that is the other model's job.
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_conv_sw_asm as G  # noqa: E402

WBASE = 1 << 41
GBASE = 1 << 42                       # GN form: the (scale, shift) table
XBASE = [(1 << 40) + d * (1 << 36) for d in range(4)]
M32, M64 = (1 << 32) - 1, (1 << 64) - 1


class Scalars:
    """the SALU subset the generated stream uses, on concrete values"""

    def __init__(self, operands, names=None):
        self.s = {}
        self.scc = 0
        self.m0 = 0
        self.op = operands            # operand name -> int (64-bit for the pointer operands)
        self.names = names or G.OPERANDS

    def val(self, tok):
        tok = tok.strip().rstrip(",")
        if tok.startswith("%"):
            return self.op[self.names[int(tok[1:])]]
        if tok.startswith("s"):
            return self.s[int(tok[1:])]
        return int(tok, 0) & M32

    def run(self, text):
        m = re.match(r"(\S+)\s+(.*)", text)
        opc, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
        if opc == "s_mov_b64":
            lo = int(re.match(r"s\[(\d+):", args[0]).group(1))
            v = self.val(args[1])
            self.s[lo], self.s[lo + 1] = v & M32, (v >> 32) & M32
            return
        dst = args[0]

        def put(v):
            if dst == "m0":
                self.m0 = v & M32
            else:
                self.s[int(dst[1:])] = v & M32

        if opc == "s_mov_b32":
            put(self.val(args[1]))
        elif opc == "s_mul_i32":
            put(self.val(args[1]) * self.val(args[2]))
        elif opc == "s_sub_u32":
            a, b = self.val(args[1]), self.val(args[2])
            put(a - b)
            self.scc = int(b > a)
        elif opc == "s_add_u32":
            v = self.val(args[1]) + self.val(args[2])
            put(v)
            self.scc = int(v > M32)
        elif opc == "s_addc_u32":
            v = self.val(args[1]) + self.val(args[2]) + self.scc
            put(v)
            self.scc = int(v > M32)
        elif opc == "s_cmp_lt_u32":
            self.scc = int(self.val(args[0]) < self.val(args[1]))
        elif opc == "s_cselect_b32":
            put(self.val(args[1]) if self.scc else self.val(args[2]))
        else:
            raise AssertionError("unmodelled scalar instruction: " + text)

    def pair(self, lo):
        return self.s[lo] | (self.s[lo + 1] << 32)


def run_stream(ops, on_op, scal):
    """sequential execution with the loop branch taken as the scalar state says"""
    labels = {o.meta["name"]: i for i, o in enumerate(ops) if o.kind == "L"}
    pc, n = 0, 0
    while pc < len(ops):
        o = ops[pc]
        n += 1
        assert n < 2_000_000
        if o.kind in ("S", "m0"):
            scal.run(o.text)
        elif o.kind == "J":
            if scal.scc:
                pc = labels[o.meta["target"]]
                continue
        elif o.kind != "L":
            on_op(o)
        pc += 1


# ------------------------------------------------------------------------------------------------------------------------
def check_schedule(nbj, cin, wave=0, up=False, f2=False, gn=False):
    c = G.Cfg(nbj, up, f2, gn)
    ops = G.generate(c)
    ncb = cin // 32
    scal = Scalars(dict(wbase=WBASE, xb0=XBASE[0], xb1=XBASE[1], xb2=XBASE[2], xb3=XBASE[3], cin2=2 * cin, nbody=cin // 64,
                        wdst=wave * 1024, hdst=wave * 1024, hdst5=min(4 + wave, 6) * 1024 if up else 20 * 1024, gbase=GBASE),
                   c.OPERANDS)
    st = dict(epoch=0, m0_fresh=False)
    reads, lg = [], []                 # every ds_read record; LDS operations still in flight, in order: ("r", read index) / ("w", fill index)
    quads = {}                         # GN form: ring position -> the halo piece it holds (load record index, transformed dwords)
    tab = []                           # GN form: the table loads, in order (fill record indices)
    dmas, vmq = [], []
    content = {("H", d): None for d in range(c.NSLOT)}
    content.update({("W", k): None for k in range(c.NS)})
    dirty = {r: False for r in content}     # a multi-step refill (halo slot) is in progress
    hpieces = {d: set() for d in range(c.NSLOT)}
    reg_sym, reg_read = {}, {}
    acc = {}
    blocks = {}                        # region -> set of 1-KiB blocks this wave's pieces covered in the current refill

    def on_op(o):
        if o.kind == "B":
            st["epoch"] += 1
        elif o.kind == "W":
            if "lgkm" in o.meta:
                while len(lg) > o.meta["lgkm"]:
                    kind, idx = lg.pop(0)
                    if kind == "r":
                        reads[idx]["retired"] = st["epoch"]
                    else:
                        dmas[idx]["landed"] = st["epoch"]
            if "vm" in o.meta:
                while len(vmq) > o.meta["vm"]:
                    dmas[vmq.pop(0)]["landed"] = st["epoch"]
        elif o.kind == "D":
            region = o.meta["region"]
            # M0: written by the instruction pair right before (run_stream executed it), decode the destination
            base = 0 if region[0] == "H" else c.W_BASE
            size = c.SLOT if region[0] == "H" else c.W_STAGE
            rel = scal.m0 - base - region[1] * size
            assert 0 <= rel < size and rel % 1024 == 0, ("M0 outside its region", o.text, scal.m0, region)
            for r in reads:                                     # WAR: every earlier read of the region retired before the last barrier
                if region in r["regions"] and region not in r["cleared"]:
                    assert r.get("retired") is not None and r["retired"] < st["epoch"], ("LDS-DMA into a region a wave may still read", o.text, r)
                    r["cleared"].add(region)
            if region[0] == "W":
                off = scal.pair(G.S_WB) - WBASE
                tap, rem = divmod(off, 2 * cin)
                assert 0 <= tap < 27 and rem % 64 == 0 and rem // 64 < ncb, ("weight pointer out of range", off)
                sym = (rem // 64, tap)
                blocks.setdefault(region, set())
                if any(d["region"] == region and d["sym"] != sym and d["landed"] is None for d in dmas):
                    raise AssertionError("two different steps in flight into one stage")
            else:
                off = scal.pair(G.S_X[region[1]]) - XBASE[region[1]]
                assert off % 64 == 0 and 0 <= off // 64 < ncb, ("halo pointer out of range", off)
                sym = (off // 64, "frame")
                k = o.meta["piece"]
                if not hpieces[region[1]]:
                    dirty[region] = True
                hpieces[region[1]].add(k)
            dmas.append(dict(region=region, sym=sym, issued=st["epoch"], landed=None, piece=o.meta.get("piece"), rel=rel // 1024))
            vmq.append(len(dmas) - 1)
            content[region] = None if region[0] == "H" else content[region]
        elif o.kind == "G" and o.meta.get("table"):
            off = scal.pair(G.S_G) - GBASE
            assert off % 256 == 0 and 0 <= off // 256 < ncb, ("table pointer out of range", off)
            if o.meta["tag"][1] == 0:
                for q in quads.values():                        # the rows are not rewritten under a transform that still needs them
                    assert q["fma"] in (0, 8), ("table load while a piece is half transformed", o.text)
                tab.append([])
            dmas.append(dict(region=None, sym=off // 256, landed=None))
            tab[-1].append(len(dmas) - 1)
            vmq.append(len(dmas) - 1)
        elif o.kind == "G":
            slot = o.meta["region"][1]
            off = scal.pair(G.S_X[slot]) - XBASE[slot]
            assert off % 64 == 0 and 0 <= off // 64 < ncb, ("halo pointer out of range", off)
            assert o.meta["ring"] not in quads, ("register quad loaded while it holds an unwritten piece", o.text)
            dmas.append(dict(region=None, sym=(off // 64, "frame"), landed=None))
            quads[o.meta["ring"]] = dict(load=len(dmas) - 1, slot=slot, piece=o.meta["piece"], fma=0, done=set(), key=o.meta["tag"][1:])
            vmq.append(len(dmas) - 1)
        elif o.kind == "V":
            q = quads[o.meta["quad"]]
            assert q["key"] == o.meta["key"], ("transform of a quad that holds another piece", o.text, q)
            if "src" in o.meta:
                assert dmas[q["load"]]["landed"] is not None, ("transform reads a piece that has not been waited for", o.text)
            if o.meta.get("tab"):
                assert tab and len(tab[-1]) == 4 and all(dmas[i]["landed"] is not None for i in tab[-1]), ("scale / shift rows in flight", o.text)
                assert dmas[tab[-1][0]]["sym"] == dmas[q["load"]]["sym"][0], ("rows of another channel block", o.text,
                                                                             dmas[tab[-1][0]]["sym"], dmas[q["load"]]["sym"])
                q["fma"] += 1
            if "dstw" in o.meta:
                q["done"].add(o.meta["dstw"])
        elif o.kind == "Wd":
            region = o.meta["region"]
            q = quads.pop(o.meta["quad"])
            assert q["done"] == {0, 1, 2, 3} and q["fma"] == 8 and q["slot"] == region[1] and q["piece"] == o.meta["piece"], (o.text, q)
            for r in reads:                                     # WAR, as for an LDS-DMA piece
                if region in r["regions"] and region not in r["cleared"]:
                    assert r.get("retired") is not None and r["retired"] < st["epoch"], ("LDS write into a region a wave may still read", o.text, r)
                    r["cleared"].add(region)
            if not hpieces[region[1]]:
                dirty[region] = True
            hpieces[region[1]].add(o.meta["piece"])
            dmas.append(dict(region=region, sym=dmas[q["load"]]["sym"], issued=st["epoch"], landed=None, piece=o.meta["piece"]))
            lg.append(("w", len(dmas) - 1))
            content[region] = None
        elif o.kind == "R":
            # the regions some wave reads with this instruction: in the two-frame form the second frame's waves read slot dt + 1
            regions = [("H", d) for d in o.meta["slots"]] if o.meta["region"][0] == "H" else [o.meta["region"]]
            for region in regions:
                pend_fill = [d for d in dmas if d["region"] == region and not d.get("done")]
                for d in pend_fill:
                    assert d["landed"] is not None and d["landed"] < st["epoch"], ("fragment read of a region whose fill is not published", o.text, d)
                if pend_fill:      # publish: the region's content is what its last complete refill carried
                    syms = {d["sym"] for d in pend_fill}
                    assert len(syms) == 1, ("mixed contents", region, syms)
                    if region[0] == "H":
                        assert hpieces[region[1]] == set(range(2 * c.NT)), ("halo slot read while its refill is incomplete", region, hpieces[region[1]])
                        hpieces[region[1]] = set()
                        dirty[region] = False
                    content[region] = syms.pop()
                    for d in pend_fill:
                        d["done"] = True
                assert not dirty[region], ("halo slot read during its refill", o.text)
            region = regions[0]
            if len(regions) == 2:      # both frames' waves must see the same channel block
                assert content[regions[0]] is None or content[regions[1]] is None or content[regions[0]][0] == content[regions[1]][0], \
                    ("the two frames read different channel blocks", o.text, content[regions[0]], content[regions[1]])
                assert (content[regions[0]] is None) == (content[regions[1]] is None), o.text
            sym = content[region]
            dst = o.meta["dst"]
            if o.meta["frag"][0] == "A":
                i = o.meta["frag"][1]
                assert dst in (c.VA[0] + 4 * i, c.VA[1] + 4 * i)
                voff = o.meta["off"] // 64
                dt, rem = divmod(voff, c.SLOTV)
                assert dt == region[1], o.text
                if up:      # two taps may read the same halo row (the upsample repeats it): the tap is the generator's claim,
                    tap = o.meta["step"] % 27                       # checked against the offset it must produce
                    assert c.a_offset(tap, i) == (o.meta["off"], o.meta["dw"]) and tap // 9 == dt, o.text
                else:
                    row, dw = divmod(rem, c.PITCH)
                    dh = row - i
                    assert 0 <= dh < 3 and 0 <= dw < 3 and dw == o.meta["dw"], o.text
                    tap = 9 * dt + 3 * dh + dw
                reg_sym[dst] = None if sym is None else ("A", sym[0], tap, i)
            else:
                j = o.meta["frag"][1]
                assert dst == c.VB + 4 * j
                reg_sym[dst] = None if sym is None else ("B", sym[0], sym[1], j)
            reads.append(dict(regions=regions, dst=dst, issued=st["epoch"], retired=None, cleared=set()))
            lg.append(("r", len(reads) - 1))
            reg_read[dst] = len(reads) - 1
        elif o.kind == "M":
            for reg in (o.meta["a"], o.meta["b"]):
                assert reads[reg_read[reg]]["retired"] is not None, ("MFMA reads a fragment whose ds_read has not been waited for", o.text)
            a, b = reg_sym[o.meta["a"]], reg_sym[o.meta["b"]]
            assert a is not None and b is not None and a[0] == "A" and b[0] == "B" and a[1:3] == b[1:3], ("mismatched operands", o.text, a, b)
            assert a[3] == o.meta["i"] and b[3] == o.meta["j"]
            key = (o.meta["j"], o.meta["i"])
            assert a[1:3] not in acc.setdefault(key, set()), ("product accumulated twice", key, a)
            acc[key].add(a[1:3])

    run_stream(ops, on_op, scal)
    want = {(cb, tap) for cb in range(ncb) for tap in range(27)}
    assert len(acc) == c.NBJ * G.NB
    for key, got in acc.items():
        assert got == want, (key, sorted(want - got)[:4], sorted(got - want)[:4])
    assert not lg and not vmq and not quads, "the loop exits with LDS reads, LDS-DMA pieces or halo pieces in flight"
    return dict(instructions=len(ops), reads=len(reads), pieces=len(dmas), barriers=st["epoch"])


def piece_coverage(nbj, up=False, f2=False):
    """the pieces the four waves issue for one refill cover every 1-KiB block of the stage / frame slot"""
    c = G.Cfg(nbj, up, f2)
    w = sorted({4 * k + wv for k in range(c.NWP) for wv in range(4)})
    h = sorted({min(4 * k + wv, c.NPIECE - 1) for k in range(2 * c.NT) for wv in range(4)})
    return w == list(range(c.W_STAGE // 1024)), h == list(range(c.NPIECE))


def up_key(ww):
    """swizzle key of source halo column ww (0..9) in the upsampled form: pairs of lanes share a voxel, so the 16-row fragment
    read touches 5-6 voxels per chunk value; 2 for ww >= 6 separates the ones four columns apart"""
    return np.where(np.asarray(ww) >= 6, 2, 0)


# ------------------------------------------------------------------------------------------------------------------------
def _gn_operands(op, wave, brick, dims):
    """GN form (plain halo geometry): a lane always fetches chunk lane % 4 of its voxel (hoff: no swizzle) and writes it to the swizzled
    LDS position itself (hdw); goff = this lane's 64 bytes (8 scales, 8 shifts) inside a channel block's 256-byte table rows"""
    hb, wb = brick
    H, W, Cin = dims
    lane = np.arange(64)
    sub, pos = lane >> 2, lane & 3
    for k in range(6):
        q = min(4 * k + wave, 20)
        v = np.minimum(16 * q + sub, 323)
        hh, ww = v // 18, v % 18
        hs = np.clip(hb * 16 - 1 + hh, 0, H - 1)
        ws = np.clip(wb * 16 - 1 + ww, 0, W - 1)
        op["hoff%d" % k] = ((hs * W + ws) * Cin + pos * 8) * 2
        op["hdw%d" % k] = v * 64 + ((pos ^ ((ww >> 1) & 3)) << 4)
    op["goff"] = pos * 64
    op["gbase"] = GBASE


def wrapper_operands_f2(wave, geom, tile, gn=False):
    """convsw2_kernel's operands: tile = (frame pair tp, brick row, brick column, 0); wave = (frame, brick half)"""
    T, H, W, Cin, Cout, wrs = geom
    tp, hb, wb, _ = tile
    lane = np.arange(64)
    q4, l15 = lane >> 4, lane & 15
    fr, half = wave >> 1, wave & 1
    sub, pos = lane >> 2, lane & 3
    op = {}
    slot = G.Cfg(8, f2=True).SLOT
    for dw in range(3):
        op["xa%d" % dw] = fr * slot + (144 * half + l15) * 64 + ((q4 ^ (((l15 + dw) >> 1) & 3)) << 4)
    op["yb"] = l15 * 64 + ((q4 ^ ((l15 >> 1) & 3)) << 4)
    for k in range(4):
        nl = 16 * (4 * (k & 1) + wave) + sub
        n = np.minimum(nl, Cout - 1)
        op["woff%d" % k] = (n * wrs + (pos ^ ((nl >> 1) & 3)) * 8) * 2
    for k in range(6):
        q = min(4 * k + wave, 20)
        v = np.minimum(16 * q + sub, 323)
        hh, ww = v // 18, v % 18
        hs = np.clip(hb * 16 - 1 + hh, 0, H - 1)
        ws = np.clip(wb * 16 - 1 + ww, 0, W - 1)
        op["hoff%d" % k] = ((hs * W + ws) * Cin + (pos ^ ((ww >> 1) & 3)) * 8) * 2
    for d in range(4):
        fs = min(max(2 * tp + d - 2, 0), T - 1)
        op["xb%d" % d] = XBASE[0] + fs * H * W * Cin * 2
    op.update(wbase=WBASE, cin2=2 * Cin, nbody=Cin // 64, wdst=wave * 1024, hdst=wave * 1024, hdst5=20 * 1024)
    if gn:
        _gn_operands(op, wave, (hb, wb), (H, W, Cin))
    return op


def wrapper_operands(nbj, wave, geom, tile, up=(False, False), gn=False):
    """per-lane / per-wave asm operands of convsw_kernel<NBJ, UP> for one tile -- the C++ formulas, re-stated.
    geom = SOURCE dims (T, H, W, Cin, Cout, wrs); up = (up_t, up_hw); tile = (output frame, brick row, brick column, n0)"""
    T, H, W, Cin, Cout, wrs = geom
    t, hb, wb, n0 = tile
    up_t, up_hw = up
    lane = np.arange(64)
    q4, l15 = lane >> 4, lane & 15
    wm, wn = wave >> 1, wave & 1
    sub, pos = lane >> 2, lane & 3
    op = {}
    for dw in range(3):
        if up_hw:
            ww = ((l15 + dw - 1) >> 1) + 1
            op["xa%d" % dw] = (40 * wm + ww) * 64 + ((q4 ^ up_key(ww)) << 4)
        else:
            op["xa%d" % dw] = (144 * wm + l15) * 64 + ((q4 ^ (((l15 + dw) >> 1) & 3)) << 4)
    op["yb"] = (wn * nbj * 16 + l15) * 64 + ((q4 ^ ((l15 >> 1) & 3)) << 4)
    for k in range(4):
        nl = 16 * (4 * (k & (nbj // 2 - 1)) + wave) + sub
        n = np.minimum(n0 + nl, Cout - 1)
        op["woff%d" % k] = (n * wrs + (pos ^ ((nl >> 1) & 3)) * 8) * 2
    for k in range(6):
        if up_hw:
            q = min(4 * (k & 1) + wave, 6)
            v = np.minimum(16 * q + sub, 99)
            hh, ww = v // 10, v % 10
            hs = np.clip(hb * 8 - 1 + hh, 0, H - 1)
            ws = np.clip(wb * 8 - 1 + ww, 0, W - 1)
            key = up_key(ww)
        else:
            q = min(4 * k + wave, 20)
            v = np.minimum(16 * q + sub, 323)
            hh, ww = v // 18, v % 18
            hs = np.clip(hb * 16 - 1 + hh, 0, H - 1)
            ws = np.clip(wb * 16 - 1 + ww, 0, W - 1)
            key = (ww >> 1) & 3
        op["hoff%d" % k] = ((hs * W + ws) * Cin + (pos ^ key) * 8) * 2
    for dt in range(3):
        tu = max(t + dt - 2, 0)
        fs = (0 if tu == 0 else 1 + ((tu - 1) >> 1)) if up_t else tu
        op["xb%d" % dt] = XBASE[0] + fs * H * W * Cin * 2          # one tensor: frame fs of batch item 0
    op["xb3"] = op["xb2"]
    op.update(wbase=WBASE, cin2=2 * Cin, nbody=Cin // 64, wdst=wave * 1024, hdst=wave * 1024,
              hdst5=(min(4 + wave, 6) if up_hw else 20) * 1024)
    if gn:
        assert not up_hw
        _gn_operands(op, wave, (hb, wb), (H, W, Cin))
    return op


def _bf16_pairs(u32):
    """[..., 4] uint32 -> [..., 8] float32 (low half first)"""
    lo = (u32.astype(np.uint32) << 16).view(np.float32)
    hi = (u32.astype(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    return np.stack([lo, hi], axis=-1).reshape(*u32.shape[:-1], -1)


def bf16_rne(f):
    """float32 array -> bf16 bits (round to nearest even; finite inputs), as v_cvt_pk_bf16_f32 / f32_to_bf16_bits"""
    u = np.asarray(f, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint32)


def gn_silu(x, scale, shift):
    """the transform the GN form applies to a conv input (csrc/groupnorm.hip::gn_apply_kernel with silu): float32 in, float32
    (bf16-exact) out; the SAME numpy operations the emulated instructions run, so the conv inputs of both sides are bit-equal"""
    y = (x.astype(np.float64) * scale.astype(np.float64) + shift.astype(np.float64)).astype(np.float32)      # v_fma_f32
    y = (bf16_rne(y) << 16).astype(np.uint32).view(np.float32)
    e = np.exp2((y * np.float32(-1.4426950408889634)).astype(np.float32)).astype(np.float32)
    r = (np.float32(1.0) / (np.float32(1.0) + e)).astype(np.float32)
    return (bf16_rne((y * r).astype(np.float32)) << 16).astype(np.uint32).view(np.float32)


def gn_table(scale, shift):
    """[Cin] scales, shifts -> the table of one batch item: [Cin / 8][16] float32 = 8 scales then 8 shifts per 8-channel chunk"""
    return np.concatenate([scale.reshape(-1, 8), shift.reshape(-1, 8)], axis=1).astype(np.float32)


def emulate_tile(nbj, x_bits, w_bits, geom, tile, up=(False, False), f2=False, gn_tab=None):
    """x_bits [T, H, W, Cin] uint16, w_bits [Cout, wrs] uint16 -> out [256 tile rows, 32 nbj channels] float32 (no bias);
    f2: the two-frame form, out [512 rows = (frame, brick row, brick column), 128 channels];
    gn_tab (gn_table()): the GN form -- the conv input is gn_silu(x)"""
    gn = gn_tab is not None
    c = G.Cfg(nbj, up[1], f2, gn)
    ops = G.generate(c)
    xb, wbts = x_bits.reshape(-1).view(np.uint8), w_bits.reshape(-1).view(np.uint8)
    gbts = np.ascontiguousarray(gn_tab, np.float32).reshape(-1).view(np.uint8) if gn else None
    lds = np.zeros(c.SMEM, np.uint8)
    waves = []
    for wv in range(4):
        opv = wrapper_operands_f2(wv, geom, tile, gn) if f2 else wrapper_operands(nbj, wv, geom, tile, up, gn)
        waves.append(dict(op=opv, scal=Scalars({k: int(v) for k, v in opv.items() if np.ndim(v) == 0}, c.OPERANDS),
                          v=np.zeros((256, 64), np.uint32), a=np.zeros((256, 64), np.float32)))
    lane = np.arange(64)

    def vsrc(wd, tok):
        tok = tok.strip().rstrip(",")
        if tok.startswith("%"):
            return np.asarray(wd["op"][c.OPERANDS[int(tok[1:])]], dtype=np.int64)
        if tok.startswith("s"):
            return np.full(64, wd["scal"].s[int(tok[1:])], np.int64)
        if tok.startswith("v"):
            return wd["v"][int(tok[1:])].astype(np.int64)
        return np.full(64, int(tok, 0) if not tok.endswith(".0") else int(np.float32(tok).view(np.uint32)), np.int64)

    def gather16(addr):
        out = np.zeros((64, 16), np.uint8)
        for ln in range(64):
            a = int(addr[ln])
            if a >= GBASE:
                out[ln] = gbts[a - GBASE: a - GBASE + 16]
            elif a >= WBASE:
                out[ln] = wbts[a - WBASE: a - WBASE + 16]
            else:
                out[ln] = xb[a - XBASE[0]: a - XBASE[0] + 16]
        return out

    f32 = lambda a: np.asarray(a, np.int64).astype(np.uint32).view(np.float32)
    u32 = lambda f: np.asarray(f, np.float32).view(np.uint32)

    def valu(wd, text):
        m = re.match(r"(\S+)\s+v(\d+), (.*)", text)
        opc, dst, src = m.group(1), int(m.group(2)), [vsrc(wd, t) for t in m.group(3).split(",")]
        if opc == "v_lshlrev_b32_e32":
            r = ((src[1] << src[0]) & M32).astype(np.uint32)
        elif opc == "v_and_b32_e32":
            r = (src[0] & src[1]).astype(np.uint32)
        elif opc == "v_fma_f32":
            r = u32((f32(src[0]).astype(np.float64) * f32(src[1]).astype(np.float64) + f32(src[2]).astype(np.float64)).astype(np.float32))
        elif opc == "v_cvt_pk_bf16_f32":
            r = (bf16_rne(f32(src[0])) | (bf16_rne(f32(src[1])) << 16)).astype(np.uint32)
        elif opc == "v_mul_f32_e32":
            r = u32((f32(src[0]) * f32(src[1])).astype(np.float32))
        elif opc == "v_add_f32_e32":
            r = u32((f32(src[0]) + f32(src[1])).astype(np.float32))
        elif opc == "v_exp_f32_e32":
            r = u32(np.exp2(f32(src[0])).astype(np.float32))
        elif opc == "v_rcp_f32_e32":
            r = u32((np.float32(1.0) / f32(src[0])).astype(np.float32))
        else:
            raise AssertionError("unmodelled vector instruction: " + text)
        wd["v"][dst] = r

    labels = {o.meta["name"]: i for i, o in enumerate(ops) if o.kind == "L"}
    pc = 0
    while pc < len(ops):
        o = ops[pc]
        if o.kind in ("S", "m0"):
            for wd in waves:
                wd["scal"].run(o.text)
        elif o.kind == "J":
            if waves[0]["scal"].scc:
                pc = labels[o.meta["target"]]
                continue
        elif o.kind == "D":
            m = re.match(r"global_load_lds_dwordx4 (\S+), s\[(\d+):", o.text)
            for wd in waves:
                addr = wd["scal"].pair(int(m.group(2))) + vsrc(wd, m.group(1))
                data = gather16(addr)
                m0 = wd["scal"].m0
                lds[m0: m0 + 1024] = data.reshape(-1)
        elif o.kind == "G":
            m = re.match(r"global_load_dwordx4 v\[(\d+):\d+\], (\S+), s\[(\d+):\d+\](?: offset:(\d+))?", o.text)
            for wd in waves:
                addr = wd["scal"].pair(int(m.group(3))) + vsrc(wd, m.group(2)) + int(m.group(4) or 0)
                words = gather16(addr).view(np.uint32)                                            # [64, 4]
                wd["v"][int(m.group(1)): int(m.group(1)) + 4] = words.T
        elif o.kind == "V":
            for wd in waves:
                valu(wd, o.text)
        elif o.kind == "Wd":
            m = re.match(r"ds_write_b128 (\S+), v\[(\d+):\d+\] offset:(\d+)", o.text)
            src = int(m.group(2))
            for wd in waves:
                addr = vsrc(wd, m.group(1)) + int(m.group(3))
                assert (addr % 16 == 0).all() and (addr + 16 <= c.HALO).all()
                data = np.ascontiguousarray(wd["v"][src: src + 4].T)                              # [64, 4] uint32
                for ln in range(64):
                    lds[int(addr[ln]): int(addr[ln]) + 16] = data[ln].view(np.uint8)
        elif o.kind == "R":
            m = re.match(r"ds_read_b128 v\[(\d+):\d+\], (\S+) offset:(\d+)", o.text)
            dst, imm = int(m.group(1)), int(m.group(3))
            for wd in waves:
                addr = vsrc(wd, m.group(2)) + imm
                assert (addr % 16 == 0).all() and (addr + 16 <= c.SMEM).all()
                words = np.stack([lds[a: a + 16].view(np.uint32) for a in addr], axis=0)      # [64, 4]
                wd["v"][dst: dst + 4] = words.T
        elif o.kind == "M":
            m = re.match(r"v_mfma_f32_16x16x32_bf16 a\[(\d+):\d+\], v\[(\d+):\d+\], v\[(\d+):\d+\]", o.text)
            d, ra, rb = int(m.group(1)), int(m.group(2)), int(m.group(3))
            for wd in waves:
                fa = _bf16_pairs(wd["v"][ra: ra + 4].T)          # [64 lanes, 8]: row l % 16, k 8 (l / 16) ..
                fb = _bf16_pairs(wd["v"][rb: rb + 4].T)
                A = np.zeros((16, 32), np.float32)
                Bm = np.zeros((32, 16), np.float32)
                for g in range(4):
                    A[:, 8 * g: 8 * g + 8] = fa[16 * g: 16 * g + 16]
                    Bm[8 * g: 8 * g + 8, :] = fb[16 * g: 16 * g + 16].T
                D = A @ Bm                                        # [m, n]
                for e in range(4):
                    wd["a"][d + e] += D[4 * (lane >> 4) + e, lane & 15]
        elif o.kind == "X":
            m = re.match(r"v_add_u32_e32 v(\d+), (\d+), (\S+)", o.text)
            if m:
                for wd in waves:
                    wd["v"][int(m.group(1))] = (int(m.group(2)) + vsrc(wd, m.group(3))).astype(np.uint32)
            else:
                m = re.match(r"v_accvgpr_write_b32 a(\d+), 0", o.text)
                if m:
                    for wd in waves:
                        wd["a"][int(m.group(1))] = 0.0
                else:
                    assert o.text.startswith("s_nop"), o.text
        pc += 1
    out = np.zeros((512 if f2 else 256, 128 if f2 else 32 * nbj), np.float32)
    for wv, wd in enumerate(waves):
        wm, wn = (wv, 0) if f2 else (wv >> 1, wv & 1)
        for j in range(nbj):
            for i in range(G.NB):
                for e in range(4):
                    rows = wm * 128 + 16 * i + (lane & 15)
                    cols = wn * 16 * nbj + 16 * j + 4 * (lane >> 4) + e
                    out[rows, cols] = wd["a"][(j * G.NB + i) * 4 + e]
    return out


def reference_tile(x, w, geom, tile, ncols, up=(False, False)):
    """float64 conv of the tile's 256 voxels (row r = brick (r >> 4, r & 15)) x ncols channels from n0: replicate spatial padding,
    causal (first-frame replicate) time padding, weights [Cout, tap-major / channel-minor]; with up: the conv sees the nearest-
    upsampled tensor (frame 0 kept single in time: unet_causal_3d_blocks.py's UpsampleCausal3D), x is the SOURCE"""
    T, H, W, Cin, Cout, wrs = geom
    t, hb, wb, n0 = tile
    up_t, up_hw = up
    Hu, Wu = (2 * H, 2 * W) if up_hw else (H, W)
    out = np.zeros((256, ncols))
    wmat = w[n0: n0 + ncols, : 27 * Cin].astype(np.float64)
    for r in range(256):
        h, ww = hb * 16 + (r >> 4), wb * 16 + (r & 15)
        cols = []
        for dt in range(3):
            for dh in range(3):
                for dw in range(3):
                    tu = max(t + dt - 2, 0)
                    ts = (0 if tu == 0 else 1 + ((tu - 1) >> 1)) if up_t else tu
                    hu = min(max(h + dh - 1, 0), Hu - 1)
                    wu = min(max(ww + dw - 1, 0), Wu - 1)
                    hs, ws = (hu >> 1, wu >> 1) if up_hw else (hu, wu)
                    cols.append(x[ts, hs, ws])
        out[r] = wmat @ np.concatenate(cols).astype(np.float64)
    return out
