#!/bin/bash
# grouped img + txt Linear launch: kernel / model parity tests, then the in-step A/B against the two-single-calls library
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/pair; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mmdit.py -q -x -m gpu -k "gemm or mmdit or forward or block" > $O/tests.log 2>&1; tail -4 $O/tests.log
grep -q "failed\|error" $O/tests.log && { echo "tests failed"; exit 0; }
for rep in 1 2; do for v in base nopair; do
  lib=""; [ $v != base ] && lib=tools/lib/libosk_gemm_nopair.so
  OSK_ALT_LIB=$lib timeout 200 python tools/step_ab.py --steps 8 --warmup 2 --no-cpu-baseline --no-b1 --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps(dict(lib='$v', ms_per_step=d['ms_per_step'], attn_ms=d['roofline']['avg_launch_ms'])))" | tee -a $O/step_ab.jsonl
done; done
