#!/bin/bash
# A/B of the 4-wave conv kernel (conv256w_kernel) against the 8-wave conv256t_kernel<256>: parity tests, then the VAE bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/conv_w4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_kernels.py -q -x -m gpu -k "conv or vae or Conv or resnet or decode or encode" > $O/tests.log 2>&1; tail -3 $O/tests.log
for w in 0 1; do
  OSK_CONV_W4=$w timeout 600 python bench.py --workload vae --steps 5 --warmup 2 --no-cpu-baseline > $O/vae_w4_$w.json 2> $O/vae_w4_$w.err
  python - $O/vae_w4_$w.json $w <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); print("W4=%s" % sys.argv[2], d["ms_per_step"], "ms", d.get("roofline"))
PY
done
