#!/bin/bash
# gemm256x_kernel with s_memtime stamps around the address set-up, the asm statement and the epilogue of every tile
# -> tools/lib/libosk_gemm_timing.so (tools/gemm_tile_timing.py reads the sums)
set -e
cd "$(dirname "$0")/.."
python -c "import open_sora_amd.build as b; b.build_lib()"
OBJ=open_sora_amd/lib/obj; mkdir -p tools/lib
# narrow: -DOSK_GEMM_NARROW_STORES = round 3's 32-byte row pieces (A/B of the quad-transposed stores);
# narrowres: -DOSK_GEMM_NARROW_RES = the residual of the gate class read in the accumulator layout's 8-byte pieces (A/B of round 5's wide residual loads)
for v in "" ${VARIANTS:-narrow narrowres}; do
  def=""; [ "$v" = narrow ] && def=-DOSK_GEMM_NARROW_STORES; [ "$v" = narrowres ] && def=-DOSK_GEMM_NARROW_RES
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -DOSK_GEMM_TILE_TIMING $def -c open_sora_amd/csrc/gemm256x.hip -o /tmp/gemm256x_timing$v.o 2>/dev/null &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lib/libosk_gemm_timing${v:+_$v}.so $(ls $OBJ/*.o | grep -v "/gemm256x.o") /tmp/gemm256x_timing$v.o &&
    echo "built tools/lib/libosk_gemm_timing${v:+_$v}.so" ) &
done
wait
