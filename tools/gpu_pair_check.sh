cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/pair2; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "gemm" > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 400 python bench.py --model 11B --steps 3 --warmup 1 --no-cpu-baseline --no-b1 --no-extra 2>/dev/null | cut -c1-260 | tee $O/b11.json
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-b1 --no-extra 2>/dev/null | cut -c1-260 | tee $O/bxl.json
