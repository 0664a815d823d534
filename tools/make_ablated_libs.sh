#!/bin/bash
# Aggressor bisect: libosk variants whose 128-wide persistent GEMM body (gemm256p_body_n128_s0.inc) lacks one instruction
# class (WRONG RESULTS; the aggressor's output is never looked at).  -> tools/lib/libosk_abl_<variant>.so
set -e
cd "$(dirname "$0")/.."
OBJ=open_sora_amd/lib/obj
mkdir -p tools/lib
for v in ${VARIANTS:-nomfma nodma nolds noaccw nobar noepi}; do
  T=/tmp/abl_$v; rm -rf $T; mkdir -p $T/csrc $T/include; cp open_sora_amd/csrc/*.h open_sora_amd/csrc/*.inc open_sora_amd/csrc/gemm256p.hip $T/csrc/; cp include/*.h $T/include/
  mkdir -p $T/a/b; cp -r $T/csrc $T/a/b/csrc; mkdir -p $T/a/include; cp include/*.h $T/a/include/ 2>/dev/null || true
  f=$T/csrc/gemm256p_body_n128_s0.inc
  case $v in
    nomfma) sed -i '/v_mfma_/d' $f ;;
    nodma)  sed -i '/global_load_lds_dwordx4/d' $f ;;
    nolds)  sed -i '/ds_read_b128/d' $f ;;
    noaccw) sed -i '/v_accvgpr_write_b32/d' $f ;;
    nobar)  sed -i '/s_barrier/d' $f ;;
    noswap) sed -i 's/const bool wide = (((uintptr_t)crow) \& 15) == 0;/const bool wide = false;/' $T/csrc/gemm_epilogue.h; grep -c "wide = false" $T/csrc/gemm_epilogue.h ;;
    noaccr) sed -i 's/  Geo::template read<T>(acc);/  for (int i_ = 0; i_ < 16; ++i_) acc[i_] = (float)(l31 + i_);/' $T/csrc/gemm_epilogue.h; grep -c "l31 + i_" $T/csrc/gemm_epilogue.h ;;
    nostore) sed -i 's/\*reinterpret_cast<uint4\*>(crow + (qd + hi) \* 8) = make_uint4(sx\[0\], sy\[0\], sx\[1\], sy\[1\]);/asm volatile("" :: "v"(sx[0]), "v"(sy[0]), "v"(sx[1]), "v"(sy[1]));/' $T/csrc/gemm_epilogue.h; grep -c 'asm volatile("" :: "v"(sx' $T/csrc/gemm_epilogue.h ;;
    noepi)  sed -i 's/epi::epilogue_all<Geo<BN>, OUT_F32>(p, m0w, n0w, l31, hi, interior, folded);/if (p.M < 0) epi::epilogue_all<Geo<BN>, OUT_F32>(p, m0w, n0w, l31, hi, interior, folded);/' $T/csrc/gemm256p.hip ;;
  esac
  # csrc files include "../../include/osk.h": give the copy the same relative layout
  L=$T/x/open_sora_amd; mkdir -p $L; rm -rf $L/csrc; cp -r $T/csrc $L/csrc; mkdir -p $T/x/include; cp include/*.h $T/x/include/
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $L/csrc/gemm256p.hip -o $T/gemm256p.o 2>/dev/null
  objs=$(ls $OBJ/*.o | grep -v "/gemm256p.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lib/libosk_abl_$v.so $objs $T/gemm256p.o
  echo "built tools/lib/libosk_abl_$v.so"
done
