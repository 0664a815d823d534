#!/bin/bash
# round-2 call 8: hd512 flash attention + VAE path re-validation + VAE bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "== pytest"; timeout 1200 python -m pytest tests/test_gpu_vae.py "tests/test_gpu_baseline_geometry.py::test_vae_shipped_widths_encode_decode_vs_oracle" "tests/test_gpu_baseline_geometry.py::test_vae_shipped_widths_tiled_vs_oracle" -q -m gpu --tb=short -p no:cacheprovider -s > $O/pytest_c8.log 2>&1; tail -25 $O/pytest_c8.log | cut -c1-200
echo "== vae bench"; timeout 600 python bench.py --workload vae --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_vae_c8.json 2> $O/bench_vae_c8.err; cat $O/bench_vae_c8.json | cut -c1-900; tail -2 $O/bench_vae_c8.err
echo "== done"
