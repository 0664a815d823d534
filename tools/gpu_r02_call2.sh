#!/bin/bash
# round-2 call 2: persistent GEMM: parity, A/B against the round-1 kernel and hipBLASLt, bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mmdit.py tests/test_gpu_fp8.py tests/test_gpu_vae.py -q -m gpu --tb=short -p no:cacheprovider -x > $O/pytest_c2.log 2>&1; tail -15 $O/pytest_c2.log
echo "== ab vendor (persistent)"; timeout 300 python tools/ab_vendor.py --out $O/ab_vendor_persist.json > $O/ab_persist.log 2>&1; cat $O/ab_persist.log | cut -c1-400
echo "== ab vendor (round-1 kernel)"; OSK_GEMM_PERSIST=0 timeout 300 python tools/ab_vendor.py --out $O/ab_vendor_r1kernel.json > $O/ab_r1.log 2>&1; cat $O/ab_r1.log | cut -c1-400
echo "== bench"; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; cat $O/bench_c2.json; tail -3 $O/bench_c2.err
echo "== done"
