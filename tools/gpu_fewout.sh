#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/fewout; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vae.py -q -x -m gpu -k "conv" > $O/tests.log 2>&1; tail -3 $O/tests.log
grep -q "failed\|error" $O/tests.log && exit 0
timeout 300 python bench.py --workload vae --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-200 | tee $O/vae.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o vae -- python bench.py --workload vae --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/vae_kernel_stats.csv && grep -i "fewout\|conv3d_kernel" "$f" | cut -c1-160; rm -rf $O/prof
