#!/bin/bash
# Which change made the packed-FP32 cross-kernel interference (profiles/r03_cross_kernel_interference.md) disappear?  Libraries of four
# points of the history, ALL built WITHOUT -packed-fp32-ops (tools/lib/hist/<commit>/ = `git archive` of that commit with the flag removed
# from its build.py; "now" = this tree + tools/lib/libosk_nopk.so).  For every (aggressor tree, victim tree) pair: the aggressor tree's
# gemm256p loop (its own tools/xproc_probe.py::aggressor_proc) runs in one process, the victim tree's GEMV probe in another; the victim's
# mismatch count out of 4000 launches is the cell.
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/r06b_pk_matrix.jsonl; : > $OUT
tree() { [ "$1" = now ] && echo $ROOT || echo $ROOT/tools/lib/hist/$1; }
for agg in ${AGGS:-fdc0901 b75dd9a dc01f6a now}; do
  for kind in ${KINDS:-gemm256p}; do
  ( cd $(tree $agg); [ $agg = now ] && export OSK_ALT_LIB=tools/lib/libosk_nopk.so
    exec python - <<PY
import sys; sys.path.insert(0, '.')
from tools import xproc_probe as X
class S:
    def is_set(self): return False
X.aggressor_proc('$kind', S())
PY
  ) > /dev/null 2>&1 &
  APID=$!
  sleep 15
  for vic in ${VICS:-fdc0901 now}; do
    ( cd $(tree $vic); [ $vic = now ] && export OSK_ALT_LIB=tools/lib/libosk_nopk.so
      timeout 300 python tools/xproc_probe.py --victim gemv --aggressor none --iters 4000 2>/dev/null | tail -1 ) > /tmp/pk_cell.json
    python - "$agg" "$kind" "$vic" <<'PY' >> $OUT
import json, sys
try:
    r = json.loads(open('/tmp/pk_cell.json').read().strip().splitlines()[-1])
    print(json.dumps({"aggressor_tree": sys.argv[1], "aggressor_kernel": sys.argv[2], "victim_tree": sys.argv[3], "victim": "gemv", "iters": r["iters"], "bad": r["bad"], "details": r.get("details", {})}))
except Exception as e:
    print(json.dumps({"aggressor_tree": sys.argv[1], "aggressor_kernel": sys.argv[2], "victim_tree": sys.argv[3], "error": str(e)}))
PY
  done
  kill $APID 2>/dev/null; wait $APID 2>/dev/null
  done
done
cat $OUT
