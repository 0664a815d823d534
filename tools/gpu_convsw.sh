#!/bin/bash
# sliding-window conv: parity tests, then layer A/B and the VAE line against the implicit-GEMM library -> gpurun_out/convsw/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/convsw; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vae.py -q -x -m gpu -k "conv" > $O/tests.log 2>&1; tail -6 $O/tests.log
if grep -q "failed\|error" $O/tests.log && [ -z "$FORCE" ]; then echo "tests failed: skipping timings"; exit 0; fi
for rep in 1 2; do for lib in "" ${AB_LIBS:-tools/lib/libosk_conv_nosw2.so tools/lib/libosk_conv_nosw.so}; do OSK_ALT_LIB=$lib timeout 200 python tools/conv_ab.py 2>/dev/null | tee -a $O/conv_ab.jsonl; done; done
for lib in "" ${AB_LIBS:-tools/lib/libosk_conv_nosw2.so tools/lib/libosk_conv_nosw.so}; do
  OSK_ALT_LIB=$lib timeout 300 python tools/step_ab.py --workload vae --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps(dict(lib='$lib' or 'shipped', ms=d['ms_per_step'], conv_ms=d['roofline'].get('total_conv_ms_per_step'), frac=d['roofline']['frac'])))" | tee -a $O/vae_ab.jsonl
done
[ -n "$SKIP_FULL" ] || { timeout 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_baseline_geometry.py -q -x -m gpu -k "not mmdit and not denoise and not xl and not XL and not sampling" > $O/tests_full.log 2>&1; tail -4 $O/tests_full.log; }
