#!/bin/bash
# In-step A/B of library switches: AB_ARMS="NAME=VAL;NAME=VAL ..." (one arm per word; ';' separates variables inside an arm),
# each arm runs bench.py (default workload) for 5 steps, interleaved twice.  Output: gpurun_out/step_ab/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/step_ab; rm -rf $O; mkdir -p $O
[ -n "$AB_TESTS" ] && { timeout 900 python -m pytest $AB_TESTS -q -x -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log; }
for rep in 1 2; do
  for arm in ${AB_ARMS:-base}; do
    ( IFS=';'; for kv in $arm; do [ "$kv" != base ] && export "$kv"; done
      timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-b1 ${BENCH_ARGS:-} 2> $O/err_${rep}.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$arm rep $rep: step %.2f ms, attention %.3f ms/launch' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))" )
  done
done | tee $O/summary.txt
