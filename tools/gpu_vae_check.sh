#!/bin/bash
# VAE-side check in one gpurun call: the VAE GPU tests (kernel level + shipped-width encode/decode vs the oracle), then the
# VAE bench line with rocprofv3 kernel stats.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/vae_check; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_vae.py tests/test_gpu_baseline_geometry.py -q -x -m gpu -k "not mmdit and not denoise and not api_fn and not xl and not XL" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 600 python bench.py --workload vae --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_vae.json 2> $O/bench_vae.err; cut -c1-330 $O/bench_vae.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o vae -- python bench.py --workload vae --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.json 2> $O/prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/vae_kernel_stats.csv && head -9 "$f" | cut -c1-150
rm -rf $O/prof
