#!/bin/bash
# round 6, GPU call 5: device-derived score bound (auto-dispatched attention pair) -- kernel tests, the whole attention / GEMM kernel files again
# (their prologues changed), model tests, the bench line with the new attention_dispatch arms
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rank_shapes.py -x -q -m gpu > $O/r06e_pytest_kernels.txt 2>&1; echo "pytest kernels rc=$?"; tail -12 $O/r06e_pytest_kernels.txt
timeout 1800 python -m pytest tests/test_gpu_mmdit.py tests/test_gpu_baseline_geometry.py -x -q -m gpu > $O/r06e_pytest_model.txt 2>&1; echo "pytest model rc=$?"; tail -5 $O/r06e_pytest_model.txt
timeout 1500 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $O/r06e_bench.json 2>$O/r06e_bench.err; echo "bench rc=$?"; tail -3 $O/r06e_bench.err
python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r06e_bench.json') if l.startswith('{')][-1])
print(r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline_gemm']['block_linear_ms_per_step'], r['roofline_gemm']['frac'], r['b1']['ms_per_step'])
for k,v in r['attention_dispatch'].items(): print(k, json.dumps(v)[:600])
PY
