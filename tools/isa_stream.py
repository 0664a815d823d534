#!/usr/bin/env python
"""Compress the gfx950 ISA of one kernel into a per-basic-block instruction-class stream, to check how the
compiler interleaved MFMAs with VALU / LDS work (no GPU needed).  Usage: isa_stream.py file.s kernel_substring
Classes: M mfma | v valu | e transcendental | p v_pk_* | a accvgpr move | d ds_read | D ds_write | g global/buffer
| s salu | w s_waitcnt(arg) | n s_nop | B s_barrier | b branch | l v_readlane/writelane"""
import re
import sys

def classify(op, args):
    if op.startswith("v_mfma"): return "M"
    if op.startswith("v_accvgpr"): return "a"
    if op in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32"): return "e"
    if op.startswith("v_pk_"): return "p"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): return "l"
    if op.startswith("v_"): return "v"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "d"
    if op.startswith("ds_"): return "D"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "g"
    if op == "s_waitcnt": return "w(" + args.replace(" ", "") + ")"
    if op == "s_nop": return "n" + args.strip()
    if op == "s_barrier": return "B"
    if op.startswith("s_cbranch") or op == "s_branch": return "b"
    if op.startswith("s_"): return "s"
    return "?" + op

def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_0-9]+:", l) and key in l and not l.startswith(".L"):
            start = i
            break
    assert start is not None, "kernel not found"
    blocks, cur, name = [], [], lines[start].rstrip(":")
    for l in lines[start + 1:]:
        if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"): break
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            blocks.append((name, cur)); cur, name = [], m.group(1); continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."): continue
        t = t.split(";")[0].strip()
        parts = t.split(None, 1)
        cur.append(classify(parts[0], parts[1] if len(parts) > 1 else ""))
    blocks.append((name, cur))
    for name, b in blocks:
        nm = sum(1 for c in b if c == "M")
        if len(sys.argv) > 3 and nm == 0: continue
        cnt = {}
        for c in b: cnt[c[0]] = cnt.get(c[0], 0) + 1
        print(f"== {name}: {len(b)} instrs, {nm} mfma, counts {cnt}")
        if nm:
            # per-MFMA gap summary
            s, gaps = "", []
            g = []
            for c in b:
                if c == "M":
                    gaps.append("".join(g)); g = []
                else:
                    g.append(c if len(c) == 1 else "[" + c + "]")
            gaps.append("".join(g))
            print("   " + " M ".join(gaps))

main()
