#!/bin/bash
# round-3 verification after the -packed-fp32-ops build: interference probes, repeatability hunt, GPU suite, bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/verify; rm -rf $O; mkdir -p $O
echo "== xproc (victim gemv vs aggressor process / stream)"
PAIRS="gemv:gemm256p gemv:attn ln:gemm256p qknorm:gemm256p" SPAIRS="gemv:gemm256p gemv:attn" ITERS=4000 bash tools/gpu_xproc.sh 2>&1 | cut -c1-200
cp gpurun_out/xproc.jsonl $O/
echo "== race hunt"
ARMS="T0 L I" RUNS=100 TAILC=400 ARM_TIMEOUT=200 bash tools/gpu_race_hunt.sh 2>&1 | grep -E "^==|rc |bad_runs" | cut -c1-300
if [ -z "$SKIP_TESTS" ]; then echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -x -m gpu --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log; fi
echo "== bench"; timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-600 $O/bench.json; tail -2 $O/bench.err
echo "== done"
