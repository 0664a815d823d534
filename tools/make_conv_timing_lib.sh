#!/bin/bash
# convsw_kernel / convsw2_kernel with s_memtime stamps around the address set-up, the asm statement (prologue + K loop) and the
# epilogue of every tile -> tools/lib/libosk_conv_timing.so (tools/conv_tile_timing.py reads the sums)
set -e
cd "$(dirname "$0")/.."
python -c "import open_sora_amd.build as b; b.build_lib()"
OBJ=open_sora_amd/lib/obj; mkdir -p tools/lib
for v in "" narrow; do    # narrow: -DOSK_CONV_NARROW_STORES = round 4's 32-byte row pieces (A/B of round 5's quad-transposed conv stores)
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -DOSK_CONV_TILE_TIMING ${v:+-DOSK_CONV_NARROW_STORES} -c open_sora_amd/csrc/conv3d_256.hip -o /tmp/conv3d_256_timing$v.o 2>/dev/null &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lib/libosk_conv_timing${v:+_$v}.so $(ls $OBJ/*.o | grep -v "/conv3d_256.o") /tmp/conv3d_256_timing$v.o &&
    echo "built tools/lib/libosk_conv_timing${v:+_$v}.so" ) &
done
wait
