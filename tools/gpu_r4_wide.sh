#!/bin/bash
# round 4: the wide attention layout -- parity, then in-step A/B against the 256-row FAST body (tools/lib/libosk_attn_narrow.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -12 | tee $O/r04e_wide_tests.txt
: > $O/r04e_wide_step.jsonl
for rep in 1 2; do for v in base narrow; do
  lib=""; [ $v != base ] && lib=tools/lib/libosk_attn_$v.so
  [ $v != base ] && [ ! -f $lib ] && continue
  OSK_ALT_LIB=$lib timeout 200 python tools/step_ab.py --steps 10 --warmup 2 --no-cpu-baseline --no-b1 --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps(dict(lib='$v', ms_per_step=d['ms_per_step'], attn_ms=r['avg_launch_ms'], frac=r['frac'])))" | tee -a $O/r04e_wide_step.jsonl
done; done
