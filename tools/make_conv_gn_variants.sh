#!/bin/bash
# A/B libraries for the GN form of the sliding-window conv (tools/conv_gn_ab.py): the shipped objects with conv3d_256.hip rebuilt
# around a re-generated GN body.  usage: tools/make_conv_gn_variants.sh name='<GN_EXP json>' ...   -> tools/lib/libosk_gn_<name>.so
set -e
cd "$(dirname "$0")/.."
python -c "import open_sora_amd.build as b; b.build_lib()"
OBJ=open_sora_amd/lib/obj; mkdir -p tools/lib
for spec in "$@"; do
  name="${spec%%=*}"; exp="${spec#*=}"
  d=/tmp/gnvar_$name; rm -rf $d; mkdir -p $d/open_sora_amd /tmp/gnvar_$name/include
  cp -r open_sora_amd/csrc $d/open_sora_amd/csrc; cp include/osk.h $d/include/
  python tools/gen_conv_sw_asm.py --out $d/open_sora_amd/csrc --gn-exp "$exp"
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -c $d/open_sora_amd/csrc/conv3d_256.hip -o $d/conv3d_256.o 2>/dev/null &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lib/libosk_gn_$name.so $(ls $OBJ/*.o | grep -v "/conv3d_256.o") $d/conv3d_256.o &&
    echo "built tools/lib/libosk_gn_$name.so" ) &
done
wait
