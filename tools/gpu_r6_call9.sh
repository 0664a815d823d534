#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu --maxfail=10 --tb=short -p no:cacheprovider -rx > $O/r06i_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/r06i_pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke > $O/r06i_smoke.txt 2>&1; tail -2 $O/r06i_smoke.txt
