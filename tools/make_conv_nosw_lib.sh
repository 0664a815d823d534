#!/bin/bash
# A/B library for the sliding-window convolution: the shipped objects with conv3d_256.hip rebuilt under -DOSK_CONV_NO_SW (every
# layer on the implicit-GEMM kernel conv256x_kernel) -> tools/lib/libosk_conv_nosw.so (use with OSK_ALT_LIB, tools/_altlib.py)
set -e
cd "$(dirname "$0")/.."
OBJ=open_sora_amd/lib/obj; mkdir -p tools/lib
for v in NO_SW NO_SW2; do   # NO_SW2: sliding window everywhere, but the Cout == 128 layers on the one-frame 128-wide form
  n=$(echo $v | tr 'A-Z' 'a-z' | tr -d _)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -DOSK_CONV_$v -c open_sora_amd/csrc/conv3d_256.hip -o /tmp/conv3d_256_$n.o 2>/dev/null &
done; wait
for n in nosw nosw2; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lib/libosk_conv_$n.so $(ls $OBJ/*.o | grep -v "/conv3d_256.o") /tmp/conv3d_256_$n.o
  echo built tools/lib/libosk_conv_$n.so
done
