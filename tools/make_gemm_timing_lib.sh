#!/bin/bash
# gemm256x_kernel with s_memtime stamps around the address set-up, the asm statement and the epilogue of every tile
# -> tools/lib/libosk_gemm_timing*.so (tools/gemm_tile_timing.py reads the sums).  Variants (A/B arms of epilogue changes):
#   (none)        the shipped epilogue
#   narrow        -DOSK_GEMM_NARROW_STORES: round 3's 32-byte row pieces (A/B of round 4's quad-transposed stores)
#   gate_serial   -DOSK_GEMM_GATE_SERIAL -DOSK_GEMM_NO_RES_TOUCH: round 4's gate epilogue (a row pair's residual loads right in front
#                 of its own conversion, no residual-line touch before the K loop)
#   gate_pipe     -DOSK_GEMM_NO_RES_TOUCH: round 5's software-pipelined residual loads without the touch
set -e
cd "$(dirname "$0")/.."
python -c "import open_sora_amd.build as b; b.build_lib()"
OBJ=open_sora_amd/lib/obj; mkdir -p tools/lib
build() {   # name, extra flags
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -DOSK_GEMM_TILE_TIMING $2 -c open_sora_amd/csrc/gemm256x.hip -o /tmp/gemm256x_timing$1.o 2>/dev/null &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lib/libosk_gemm_timing$1.so $(ls $OBJ/*.o | grep -v "/gemm256x.o") /tmp/gemm256x_timing$1.o &&
    echo "built tools/lib/libosk_gemm_timing$1.so"
}
build "" "" &
build _narrow "-DOSK_GEMM_NARROW_STORES" &
build _gate_serial "-DOSK_GEMM_GATE_SERIAL -DOSK_GEMM_NO_RES_TOUCH" &
build _gate_pipe "-DOSK_GEMM_NO_RES_TOUCH" &
wait
