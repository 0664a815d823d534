#!/bin/bash
# pytest -m gpu, the default bench line (with its vae / 11b sub-objects), optionally kernel stats -> gpurun_out/r3/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r3; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then echo "== pytest -m gpu ${PYTEST_ARGS}"; timeout 1500 python -m pytest tests -q -x -m gpu --tb=short -p no:cacheprovider ${PYTEST_ARGS} > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log; fi
if [ -z "$SKIP_BENCH" ]; then echo "== bench"; timeout 900 python bench.py ${BENCH_ARGS} > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r3/bench.json"))
def brief(x): return {k: x[k] for k in ("ms_per_step","value","step_mfma_frac") if k in x} | ({"roofline": {k: x["roofline"][k] for k in ("kernel","achieved","frac","avg_launch_ms","total_conv_ms_per_step") if k in x["roofline"]}} if x.get("roofline") else {})
print("main", brief(d), d.get("timed"))
for k in ("b1","vae","11b"):
    if k in d: print(k, brief(d[k]) if k!="b1" else d[k])
print("cpu", {k: d["cpu_baseline"][k] for k in ("value","cores","kind")} if "cpu_baseline" in d else None)
PY
tail -2 $O/bench.err; fi
if [ -n "$PROF" ]; then for w in dit vae; do echo "== rocprofv3 kernel stats: $w"; rm -rf $O/prof_$w
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o $w -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-b1 --no-extra > $O/prof_$w.json 2> $O/prof_$w.err
  f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${w}_kernel_stats.csv && head -9 "$f" | cut -c1-170; rm -rf $O/prof_$w; done; fi
echo "== done"
