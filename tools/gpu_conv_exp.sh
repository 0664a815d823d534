#!/bin/bash
# timing-only decomposition of a sliding-window conv tile (tools/make_conv_exp_libs.sh) -> gpurun_out/conv_exp.jsonl
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; : > $O/conv_exp.jsonl
export CONV_AB_SHAPES=${CONV_AB_SHAPES:-0,2,3}
for lib in "" tools/lib/libosk_convexp_noepi.so tools/lib/libosk_convexp_onebody.so tools/lib/libosk_convexp_both.so; do
  OSK_ALT_LIB=$lib timeout 200 python tools/conv_ab.py 2>/dev/null | tee -a $O/conv_exp.jsonl
done
