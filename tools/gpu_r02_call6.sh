#!/bin/bash
# round-2 call 6: L2 software prefetch in the 4-wave GEMM
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "== pytest gemm (PF=1)"; timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -x -k "gemm" > $O/pytest_c6.log 2>&1; tail -6 $O/pytest_c6.log
for cfg in "PF=0" "PF=1"; do
  echo "== epilogue bench OSK_GEMM_$cfg"; env OSK_GEMM_$cfg timeout 200 python tools/gemm_epi_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/epi_$cfg.jsonl
  echo "== ab vendor OSK_GEMM_$cfg"; env OSK_GEMM_$cfg timeout 300 python tools/ab_vendor.py --out $O/ab_vendor_$cfg.json 2>&1 | grep -v amdgpu.ids | grep gemm | cut -c1-300
done
PROF_TAG=r02e_pf1 bash tools/gpu_prof_step.sh 2>&1 | head -4
OSK_GEMM_PF=0 PROF_TAG=r02e_pf0 bash tools/gpu_prof_step.sh 2>&1 | head -4
echo "== done"
