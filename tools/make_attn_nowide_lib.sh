#!/bin/bash
# A/B library for the wide attention layout: the shipped objects, attention_fwd.hip rebuilt with -DOSK_ATTN_NO_WIDE (every bounded
# head_dim-72 call takes the 256-row FAST body)  -> tools/lib/libosk_attn_narrow.so
set -e
cd "$(dirname "$0")/.."
python -c "import open_sora_amd.build as b; b.build_lib()"
OBJ=open_sora_amd/lib/obj; mkdir -p tools/lib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -DOSK_ATTN_NO_WIDE -c open_sora_amd/csrc/attention_fwd.hip -o /tmp/attention_fwd_nowide.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lib/libosk_attn_narrow.so $(ls $OBJ/*.o | grep -v "/attention_fwd.o") /tmp/attention_fwd_nowide.o
echo "built tools/lib/libosk_attn_narrow.so"
