#!/bin/bash
# round-2 call 1: new parity tests at the BASELINE geometries, hipGraph capture, vendor A/B, bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu.txt; nproc >> $O/gpu.txt; free -g | head -2 >> $O/gpu.txt; python -c "import torch; print(torch.cuda.device_count())" >> $O/gpu.txt
echo "== pytest (new tests)"; timeout 900 python -m pytest tests/test_gpu_baseline_geometry.py tests/test_gpu_seqpar_nccl.py "tests/test_gpu_mmdit.py::test_denoise_step_is_hipgraph_capturable" -q -m gpu --tb=short -p no:cacheprovider -s -k "not full_bench_length" > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
echo "== ab vendor"; bash tools/gpu_ab_vendor.sh > $O/ab_vendor.log 2>&1; tail -16 $O/ab_vendor.log
echo "== bench"; timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err
echo "== done"
