#!/bin/bash
# the final tree of round 6: the driver's three steps (GPU suite, smoke, default bench line)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/r06j_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/r06j_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06j_smoke.txt 2>&1; tail -1 $O/r06j_smoke.txt
/usr/bin/time -v timeout 1500 python bench.py > $O/r06j_bench.json 2> $O/r06j_bench.err; echo "bench rc=$?"; grep "Elapsed (wall" $O/r06j_bench.err; cut -c1-300 $O/r06j_bench.json
