#!/bin/bash
# A/B libraries for attention schedule experiments: scratch copy of csrc, generator options, only the attention objects rebuilt.
#   VARIANTS="name|generator args;name2|args2" bash tools/make_attn_variants.sh   -> tools/lib/libosk_attn_<name>.so
set -e
cd "$(dirname "$0")/.."
OBJ=open_sora_amd/lib/obj
mkdir -p tools/lib
IFS=';' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do
  name=${v%%|*}; gargs=${v#*|}
  T=/tmp/attnvar_$name; rm -rf $T; mkdir -p $T/x/open_sora_amd $T/x/include; cp -r open_sora_amd/csrc $T/x/open_sora_amd/csrc; cp include/*.h $T/x/include/
  python tools/gen_attn_asm.py --out $T/x/open_sora_amd/csrc $gargs
  objs=$(ls $OBJ/*.o | grep -v "/attention_asm72.o\|/attention_asm128.o\|/attention_asm72w.o")
  for f in attention_asm72 attention_asm128 attention_asm72w; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -c $T/x/open_sora_amd/csrc/$f.hip -o $T/$f.o 2>/dev/null &
  done; wait
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lib/libosk_attn_$name.so $objs $T/attention_asm72.o $T/attention_asm128.o $T/attention_asm72w.o
  echo "built tools/lib/libosk_attn_$name.so  ($gargs)"
done
