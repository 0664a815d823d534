#!/bin/bash
# A/B of the conv kernels for Cout >= 256: parity tests per arm, then the VAE bench line.
#   W4=0 : 8-wave conv256t_kernel<256>    W4=1 : 4-wave conv256w_kernel    W4=1 PF=1 : + L2 prefetch of the activation lines
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/conv_w4; mkdir -p $O
for arm in "0 0" "1 0" "1 1"; do
  set -- $arm
  export OSK_CONV_W4=$1 OSK_CONV_PF=$2
  if [ "$arm" != "0 0" ]; then
    timeout 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_kernels.py -q -x -m gpu -k "conv or vae or Conv or resnet or decode or encode" > $O/tests_$1$2.log 2>&1; tail -1 $O/tests_$1$2.log
  fi
  timeout 600 python bench.py --workload vae --steps 5 --warmup 2 --no-cpu-baseline > $O/vae_$1$2.json 2> $O/vae_$1$2.err
  python - $O/vae_$1$2.json "$arm" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); print("W4 PF = %s" % sys.argv[2], d["ms_per_step"], "ms; conv", d["roofline"]["total_conv_ms_per_step"], "ms", d["roofline"]["achieved"], "TF/s")
PY
done
