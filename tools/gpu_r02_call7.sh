#!/bin/bash
# round-2 call 7: new tests (seqpar on HIP with staged collectives, ckpt, blend, tiled VAE) + full-length block parity
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "== pytest"; timeout 1200 python -m pytest tests/test_gpu_seqpar_1gpu.py tests/test_gpu_vae.py "tests/test_gpu_mmdit.py::test_checkpoint_load_and_rope_convention_on_gpu" -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_c7.log 2>&1; tail -25 $O/pytest_c7.log
echo "== full-length blocks"; timeout 900 python -m pytest "tests/test_gpu_baseline_geometry.py::test_xl_blocks_at_full_bench_length" -q -m gpu --tb=short -p no:cacheprovider -s > $O/pytest_c7b.log 2>&1; tail -8 $O/pytest_c7b.log
echo "== done"
