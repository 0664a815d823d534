#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" > $O/r06f_pytest_gemm.txt 2>&1; echo "pytest gemm rc=$?"; tail -3 $O/r06f_pytest_gemm.txt
for i in 1 2 3; do
  for lib in shipped scalar_gelu; do
    if [ $lib = scalar_gelu ]; then export OSK_ALT_LIB=tools/lib/libosk_gemm_scalar_gelu.so; else unset OSK_ALT_LIB; fi
    timeout 600 python tools/step_ab.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-b1 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(json.dumps({'lib':'$lib','run':$i,'ms_per_step':r['ms_per_step'],'attn_ms':r['roofline']['avg_launch_ms'],'gemm_ms_per_step':r['roofline_gemm']['block_linear_ms_per_step'],'gemm_frac':r['roofline_gemm']['frac']}))" >> $O/r06f_gelu_packed_ab.jsonl
  done
done
unset OSK_ALT_LIB
cat $O/r06f_gelu_packed_ab.jsonl
timeout 2400 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_seqpar_1gpu.py tests/test_gpu_vae.py tests/test_gpu_fp8.py tests/test_gpu_stdit_shapes.py -x -q -m gpu > $O/r06f_pytest_rest.txt 2>&1; echo "pytest rest rc=$?"; tail -5 $O/r06f_pytest_rest.txt
timeout 600 python bench.py --workload vae --steps 10 --warmup 3 --no-cpu-baseline > $O/r06f_bench_vae.json 2>/dev/null; cut -c1-330 $O/r06f_bench_vae.json
PROF_TAG=r06f_vae BENCH_ARGS="--workload vae" bash tools/gpu_prof_step.sh 2>&1 | tail -16
