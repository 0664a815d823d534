#!/bin/bash
# last code state of round 6 (packed GELU withdrawn): GEMM / overlap regression, PMC traffic of the step with the V^T kernel counted, the default bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_overlap.py -x -q -m gpu -k "gemm or packed or victim" > $O/r06h_pytest_gemm_overlap.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r06h_pytest_gemm_overlap.txt
PMC_ONLY_DIT=1 bash tools/gpu_pmc_kernels.sh > $O/r06h_pmc_dit.log 2>&1; cp gpurun_out/pmc_gemm_conv/summary.txt $O/r06h_pmc_dit_summary.txt; cat $O/r06h_pmc_dit_summary.txt
timeout 1500 python bench.py > $O/r06h_bench.json 2> $O/r06h_bench.err; echo "bench rc=$?"; cut -c1-400 $O/r06h_bench.json
