#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
S=$(date +%s); timeout 1500 python bench.py > $O/r06j_bench.json 2> $O/r06j_bench.err; echo "bench rc=$? wall $(( $(date +%s) - S )) s"; cut -c1-300 $O/r06j_bench.json
