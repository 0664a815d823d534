#!/bin/bash
# The builder-run bench lines beside the default one: XL fp8 mode, 11B geometry in bf16 and fp8 mode.  Output: gpurun_out/matrix/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/matrix; rm -rf $O; mkdir -p $O
run() { name=$1; shift; timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-b1 "$@" > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("%-10s step %.1f ms, %.0f TFLOP/s; %s %.0f TFLOP/s (%.3f)" % (sys.argv[2], d["ms_per_step"], d["step_tflops"], r["kernel"], r["achieved"], r["frac"]))
PY
}
run xl_fp8 --fp8
run 11b --model 11B
run 11b_fp8 --model 11B --fp8
