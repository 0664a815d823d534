#!/bin/bash
# gemm256x.hip with / without the -packed-fp32-ops flag (scalar GELU in both): is the compiler's own packing in this one-wave-per-SIMD kernel worth an exception?
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for i in 1 2 3; do
  for lib in scalar_gelu scalar_gelu_noflag; do
    export OSK_ALT_LIB=tools/lib/libosk_gemm_$lib.so
    timeout 600 python tools/step_ab.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-b1 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(json.dumps({'lib':'$lib','run':$i,'ms_per_step':r['ms_per_step'],'attn_ms':r['roofline']['avg_launch_ms'],'gemm_ms_per_step':r['roofline_gemm']['block_linear_ms_per_step'],'gemm_frac':r['roofline_gemm']['frac']}))" >> $O/r06g_gemm256x_flag_ab.jsonl
  done
done
unset OSK_ALT_LIB
cat $O/r06g_gemm256x_flag_ab.jsonl
