#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for v in "" _notrim; do
  OSK_ALT_LIB=tools/lib/libosk_gemm_timing$v.so timeout 600 python tools/gemm_tile_timing.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l)
    if 'ticks_per_tile' in r: print(json.dumps({'lib':'timing$v','what':r['what'],'epilogue':r['epilogue'],'ms':r['ms_per_launch'],**r['ticks_per_tile']}))" >> $O/r06d_gemm_setup_ab.jsonl
done
cat $O/r06d_gemm_setup_ab.jsonl
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pkfp32_repro.hip -o /tmp/pkfp32_repro -ldl 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops tools/pkfp32_repro.hip -o /tmp/pkfp32_repro_nopk -ldl 2>/dev/null
for arm in "/tmp/pkfp32_repro open_sora_amd/lib/libosk_hip.so" "/tmp/pkfp32_repro_nopk open_sora_amd/lib/libosk_hip.so" "/tmp/pkfp32_repro none"; do
  echo "{\"cmd\": \"$arm 4000\", \"result\": $(timeout 300 $arm 4000 | tail -1)}" >> $O/r06d_pkfp32_repro.jsonl
done
cat $O/r06d_pkfp32_repro.jsonl
for i in 1 2; do
  for lib in shipped narrowres; do
    if [ $lib = narrowres ]; then export OSK_ALT_LIB=tools/lib/libosk_conv_narrowres.so; else unset OSK_ALT_LIB; fi
    timeout 600 python tools/run_with_lib.py bench.py --workload vae --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(json.dumps({'lib':'$lib','run':$i,'ms_per_step':r['ms_per_step'],'conv_ms':r['roofline']['total_conv_ms_per_step'],'gn_fold_ms':r['gn_fold']['ms_per_step']}))" >> $O/r06d_conv_res_ab.jsonl
  done
done
unset OSK_ALT_LIB
cat $O/r06d_conv_res_ab.jsonl
timeout 600 python tools/gemm_group_ab.py > $O/r06d_gemm_group_ab.jsonl 2>&1; cat $O/r06d_gemm_group_ab.jsonl
