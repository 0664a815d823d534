#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "== pytest vae"; timeout 900 python -m pytest tests/test_gpu_vae.py "tests/test_gpu_baseline_geometry.py::test_vae_shipped_widths_encode_decode_vs_oracle" -q -m gpu --tb=short -p no:cacheprovider -x > $O/pytest_c15.log 2>&1; tail -5 $O/pytest_c15.log
for b in 0 1; do
  echo "== vae bench OSK_CONV_BRICK=$b"; OSK_CONV_BRICK=$b timeout 300 python bench.py --workload vae --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'], d['roofline']['total_conv_ms_per_step'])"
done
echo "== done"
