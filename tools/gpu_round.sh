#!/bin/bash
# One gpurun call; steps selected by $STEPS (default: all).  Everything comes back under gpurun_out/.
#   STEPS="tests micro" ATTN_VARIANTS="0 1 9" bash tools/gpu_round.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
STEPS="${STEPS:-probes tests smoke micro bench prof}"
has() { [[ " $STEPS " == *" $1 "* ]]; }
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu.txt; nproc >> $O/gpu.txt; free -g | head -2 >> $O/gpu.txt
if has probes; then echo "== probes"; (hipcc --offload-arch=gfx950 -O2 -o /tmp/probes tools/probes.hip && timeout 60 /tmp/probes) > $O/probes.txt 2>&1; grep -c PASS $O/probes.txt; fi
if has tests; then echo "== pytest gpu"; timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider ${PYTEST_ARGS:-} > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
  for v in ${ATTN_TEST_VARIANTS:-}; do echo "== pytest attention variant $v"; OSK_ATTN_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k attention --tb=short -p no:cacheprovider > $O/pytest_attn_v$v.log 2>&1; tail -4 $O/pytest_attn_v$v.log; done
fi
if has smoke; then echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log; fi
if has micro; then
  : > $O/microbench.jsonl
  for v in ${ATTN_VARIANTS:-0}; do for g in ${GEMM_VARIANTS:-0}; do
    echo "== microbench attn=$v gemm=$g"; OSK_ATTN_VARIANT=$v OSK_GEMM_VARIANT=$g timeout 300 python tools/microbench.py ${MICRO_ARGS:---quick} 2>> $O/microbench.err | sed "s/^{/{\"attn_variant\": $v, \"gemm_variant\": $g, /" | tee -a $O/microbench.jsonl
  done; done
fi
if has bench; then echo "== bench"; timeout 600 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err; fi
if has prof; then echo "== rocprof"; rm -rf $O/prof; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ${PROF_NAME:-prof} -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/rocprof_bench.json 2> $O/rocprof.err; tail -2 $O/rocprof.err; head -8 $O/prof/*kernel_stats.csv | cut -c1-160; fi
echo "== done"
