#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "== pytest mmdit"; timeout 900 python -m pytest tests/test_gpu_mmdit.py tests/test_gpu_seqpar_1gpu.py -q -m gpu --tb=short -p no:cacheprovider -x > $O/pytest_c14.log 2>&1; tail -6 $O/pytest_c14.log
echo "== bench under torchrun, 1 rank (nccl init, barrier, all_reduce path)"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-b1 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400
echo "== bench --gpus 2 self-spawn on a 1-GPU box must fail cleanly"
timeout 120 python bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline > $O/spawn2.log 2>&1; echo "rc=$?"; tail -3 $O/spawn2.log | cut -c1-200
echo "== done"
