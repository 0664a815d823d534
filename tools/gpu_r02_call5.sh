#!/bin/bash
# round-2 call 5: 4-wave GEMM (gemm256w): parity, epilogue bench and vendor A/B vs the 8-wave persistent kernel, in-step stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "== pytest gemm"; timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -x -k "gemm" > $O/pytest_c5.log 2>&1; tail -6 $O/pytest_c5.log
for cfg in "W4=0" "W4=1"; do
  echo "== epilogue bench OSK_GEMM_$cfg"; env OSK_GEMM_$cfg timeout 200 python tools/gemm_epi_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/epi_$cfg.jsonl
done
echo "== ab vendor W4=1"; timeout 300 python tools/ab_vendor.py --out $O/ab_vendor_W4.json 2>&1 | grep -v amdgpu.ids | grep gemm | cut -c1-330
echo "== mmdit tests"; timeout 600 python -m pytest tests/test_gpu_mmdit.py tests/test_gpu_fp8.py -q -m gpu --tb=short -p no:cacheprovider -x > $O/pytest_c5b.log 2>&1; tail -4 $O/pytest_c5b.log
PROF_TAG=r02d_w4 bash tools/gpu_prof_step.sh 2>&1 | head -8
OSK_GEMM_W4=0 PROF_TAG=r02d_p8 bash tools/gpu_prof_step.sh 2>&1 | head -8
echo "== done"
