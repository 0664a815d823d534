#!/bin/bash
# round 6, GPU call 2: osk_gemm_group_bf16 (V^T written by the projection) parity + model-level tests, step profile, the packed-FP32 history
# matrix, and the step on a build WITHOUT -packed-fp32-ops (A/B)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" > $O/r06b_pytest_gemm.txt 2>&1; echo "pytest gemm rc=$?"; tail -12 $O/r06b_pytest_gemm.txt
timeout 2400 python -m pytest tests/test_gpu_mmdit.py tests/test_gpu_baseline_geometry.py tests/test_gpu_fp8.py -x -q -m gpu > $O/r06b_pytest_model.txt 2>&1; echo "pytest model rc=$?"; tail -8 $O/r06b_pytest_model.txt
PROF_TAG=r06b_step BENCH_ARGS=--no-extra bash tools/gpu_prof_step.sh 2>&1 | tail -16
for i in 1 2; do
  for lib in shipped nopk; do
    if [ $lib = nopk ]; then export OSK_ALT_LIB=tools/lib/libosk_nopk.so; else unset OSK_ALT_LIB; fi
    timeout 600 python tools/step_ab.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(json.dumps({'lib':'$lib','run':$i,'ms_per_step':r['ms_per_step'],'attn_ms':r['roofline']['avg_launch_ms'],'gemm_ms_per_step':r['roofline_gemm']['block_linear_ms_per_step'],'b1_ms':r['b1']['ms_per_step']}))" >> $O/r06b_step_ab_nopk.jsonl
  done
done
unset OSK_ALT_LIB
cat $O/r06b_step_ab_nopk.jsonl
timeout 1200 bash tools/pk_matrix.sh 2>&1 | tail -12
