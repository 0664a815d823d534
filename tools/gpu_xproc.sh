#!/bin/bash
# cross-process / cross-stream interference matrix (tools/xproc_probe.py) -> gpurun_out/xproc.jsonl
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O; : > $O/xproc.jsonl
ITERS=${ITERS:-4000}
for pair in ${PAIRS:-gemv:none gemv:gemm256p gemv:gemm_small gemv:attn gemv:matmul gemv:gemv gemm256p:gemm256p gemm_small:gemm256p attn:gemm256p ln:gemm256p qknorm:gemm256p}; do
  v=${pair%%:*}; a=${pair##*:}
  timeout 240 python tools/xproc_probe.py --victim $v --aggressor $a --iters $ITERS 2>/dev/null | grep '^{' | tee -a $O/xproc.jsonl | cut -c1-600
done
for pair in ${SPAIRS:-gemv:gemm256p gemv:gemv gemm256p:gemm256p}; do
  v=${pair%%:*}; a=${pair##*:}
  timeout 240 python tools/xproc_probe.py --victim $v --aggressor none --stream-aggressor $a --iters $ITERS 2>/dev/null | grep '^{' | tee -a $O/xproc.jsonl | cut -c1-600
done
echo "== done"
