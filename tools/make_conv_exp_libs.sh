#!/bin/bash
# TIMING-ONLY experiment libraries of the sliding-window conv (wrong results): where does a tile's time go?
#   noepi = no epilogue, onebody = K loop cut to one body (2 channel blocks) whatever Cin, both = neither
set -e
cd "$(dirname "$0")/.."
OBJ=open_sora_amd/lib/obj; mkdir -p tools/lib
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops"
hipcc $F -DOSK_CONV_EXP_NOEPI -c open_sora_amd/csrc/conv3d_256.hip -o /tmp/c256_noepi.o 2>/dev/null &
hipcc $F -DOSK_CONV_EXP_ONEBODY -c open_sora_amd/csrc/conv3d_256.hip -o /tmp/c256_onebody.o 2>/dev/null &
hipcc $F -DOSK_CONV_EXP_NOEPI -DOSK_CONV_EXP_ONEBODY -c open_sora_amd/csrc/conv3d_256.hip -o /tmp/c256_both.o 2>/dev/null &
wait
for n in noepi onebody both; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lib/libosk_convexp_$n.so $(ls $OBJ/*.o | grep -v "/conv3d_256.o") /tmp/c256_$n.o
done
echo built tools/lib/libosk_convexp_{noepi,onebody,both}.so
