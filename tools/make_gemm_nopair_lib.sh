#!/bin/bash
# A/B library: osk_gemm_bf16_pair always as its two single calls (-DOSK_GEMM_NO_PAIR) -> tools/lib/libosk_gemm_nopair.so
set -e
cd "$(dirname "$0")/.."
OBJ=open_sora_amd/lib/obj; mkdir -p tools/lib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -DOSK_GEMM_NO_PAIR -c open_sora_amd/csrc/gemm_bf16.hip -o /tmp/gemm_bf16_nopair.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lib/libosk_gemm_nopair.so $(ls $OBJ/*.o | grep -v "/gemm_bf16.o") /tmp/gemm_bf16_nopair.o
echo built tools/lib/libosk_gemm_nopair.so
