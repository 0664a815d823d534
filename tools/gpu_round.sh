#!/bin/bash
# One gpurun call: probes, GPU parity tests, smoke, bench, per-kernel microbench, rocprof summary.
# Everything the session needs comes back under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu.txt
nproc >> $O/gpu.txt; free -g | head -2 >> $O/gpu.txt
echo "== probes" ; (hipcc --offload-arch=gfx950 -O2 -o /tmp/probes tools/probes.hip && timeout 60 /tmp/probes) > $O/probes.txt 2>&1; tail -3 $O/probes.txt
echo "== pytest gpu"; timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider ${PYTEST_ARGS:-} > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
if [ -z "$SKIP_BENCH" ]; then
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
echo "== microbench"; timeout 300 python tools/microbench.py ${MICRO_ARGS:-} > $O/microbench.jsonl 2> $O/microbench.err; cat $O/microbench.jsonl
echo "== microbench gemm variant 1"; OSK_GEMM_VARIANT=1 timeout 200 python tools/microbench.py --quick --gemm-only > $O/microbench_v1.jsonl 2>> $O/microbench.err; grep gemm $O/microbench_v1.jsonl | head -8
echo "== bench"; timeout 600 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err
echo "== rocprof"; rm -rf $O/prof; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r01 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/rocprof_bench.json 2> $O/rocprof.err; tail -2 $O/rocprof.err
find $O/prof -name "*stats*" | head; 
fi
echo "== done"
