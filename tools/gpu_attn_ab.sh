#!/bin/bash
# A/B of attention schedule variants (tools/make_attn_variants.sh): the bounded microbench line per library, interleaved twice
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; : > $O/attn_ab.jsonl
for rep in 1 2; do for v in base ${ABV}; do
  lib=""; [ $v != base ] && lib=tools/lib/libosk_attn_$v.so
  OSK_ALT_LIB=$lib timeout 200 python tools/microbench.py --quick --attn-only 2>/dev/null | grep -E "bound" | grep -E "xl.cfg2.b3|11b" | sed "s/^{/{\"lib\": \"$v\", /" | cut -c1-230 | tee -a $O/attn_ab.jsonl
done; done
