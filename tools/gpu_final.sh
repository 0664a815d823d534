#!/bin/bash
# Round-end validation in one gpurun call: the GPU test suite, smoke(), both bench lines, the self-spawned 2-rank line's
# refusal on a 1-GPU box, and rocprofv3 kernel stats of both workloads.  Output: gpurun_out/final/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/final; rm -rf $O; mkdir -p $O
echo "== pytest -m gpu"; timeout ${PYTEST_TIMEOUT:-1800} python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
echo "== bench (default line)"; timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json; tail -2 $O/bench.err
echo "== bench vae"; timeout 600 python bench.py --workload vae --steps 5 --warmup 2 > $O/bench_vae.json 2> $O/bench_vae.err; cut -c1-300 $O/bench_vae.json
if [ -n "$FINAL_FP8" ]; then echo "== bench fp8"; timeout 600 python bench.py --fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-b1 > $O/bench_fp8.json 2> $O/bench_fp8.err; cut -c1-300 $O/bench_fp8.json; fi
for w in dit vae; do
  echo "== rocprofv3 kernel stats: $w"
  rm -rf $O/prof_$w
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o $w -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-b1 > $O/prof_$w.json 2> $O/prof_$w.err
  f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/${w}_kernel_stats.csv && head -9 "$f" | cut -c1-170
  rm -rf $O/prof_$w
done
echo "== done"
