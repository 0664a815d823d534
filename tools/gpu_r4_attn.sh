#!/bin/bash
# round 4: FAST attention body with loader events -- parity of every attention test, then the step with the shipped library
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -15 | tee $O/r04_attn_tests.txt
for rep in 1 2; do
  timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-b1 --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps(dict(ms_per_step=d['ms_per_step'], attn_ms=r['avg_launch_ms'], frac=r['frac'])))" | tee -a $O/r04_attn_step.jsonl
done
