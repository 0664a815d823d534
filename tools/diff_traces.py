#!/usr/bin/env python
"""Compare two C-ABI call traces (OSK_TRACE=1, `json.dump(open_sora_amd._C.TRACE_LOG, f)`): first position where the entry point,
a scalar argument, a buffer's identity (order of first appearance) or an alignment differs, then a per-entry-point count table.
    python tools/diff_traces.py a.json b.json"""
import collections
import json
import sys

a, b = (json.load(open(p)) for p in sys.argv[1:3])
print(f"{len(a)} vs {len(b)} calls")
for i, (x, y) in enumerate(zip(a, b)):
    if x != y:
        print(f"first difference at call {i}:\n  A {x}\n  B {y}")
        print("  arguments that differ:", [(k, u, v) for k, (u, v) in enumerate(zip(x, y)) if u != v])
        break
else:
    print("common prefix identical")
ca, cb = collections.Counter(x[0] for x in a), collections.Counter(y[0] for y in b)
for name in sorted(set(ca) | set(cb)):
    if ca[name] != cb[name]:
        print(f"  {name}: {ca[name]} vs {cb[name]} calls")
