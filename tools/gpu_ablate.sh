#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; : > $O/ablate.jsonl
for v in ${ABL:-full nomfma nodma nolds noaccw nobar noepi}; do
  lib=""; [ $v != full ] && lib=tools/lib/libosk_abl_$v.so
  OSK_PROBE_LIB=$lib timeout 120 python tools/interfere_probe.py --victims=victim_gemvvar3,victim_gemvvar0 --aggressors=gemm256p 2>/dev/null | sed "s/^{/{\"lib\": \"$v\", /" | cut -c1-200 | tee -a $O/ablate.jsonl
done
