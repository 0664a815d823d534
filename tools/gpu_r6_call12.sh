#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_seqpar_1gpu.py -x -q -m gpu > $O/r06l_pytest_sp.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/r06l_pytest_sp.txt
