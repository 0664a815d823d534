#!/bin/bash
# in-step A/B of experiment libraries (tools/make_attn_variants.sh): ms per step + attention launch time, interleaved twice
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; : > $O/step_ab.jsonl
for rep in 1 2; do for v in base ${ABV}; do
  lib=""; [ $v != base ] && lib=tools/lib/libosk_attn_$v.so
  OSK_ALT_LIB=$lib timeout 200 python tools/step_ab.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-b1 --no-extra ${BARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps(dict(lib='$v', ms_per_step=d['ms_per_step'], attn_ms=r['avg_launch_ms'], attn_tflops=r['achieved'], frac=r['frac'])))" | tee -a $O/step_ab.jsonl
done; done
