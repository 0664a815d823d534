#!/bin/bash
# round 6, GPU call 1: new parity tests (rank shapes, VAE plug-in on device), packed-FP32 re-probe on a no-flag build, the bench line with
# the new sub-objects, world-1 dry run of the first-contact kit
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rank_shapes.py "tests/test_gpu_vae.py::test_plugin_targets_on_device" -x -q -m gpu > $O/r06a_pytest_new.txt 2>&1
echo "pytest new rc=$?"; tail -15 $O/r06a_pytest_new.txt
# packed-FP32 interference on today's tree, library built WITHOUT -packed-fp32-ops (tools/make_nopk_lib.sh), then the shipped build as control
for agg in gemm256p attn gemm256x; do
  OSK_ALT_LIB=tools/lib/libosk_nopk.so timeout 600 python tools/xproc_probe.py --victim gemv --aggressor $agg --iters 4000 2>/dev/null | tail -1 | sed 's/^/{"lib": "nopk", "r": /; s/$/}/' >> $O/r06a_xproc_nopk.jsonl
done
OSK_ALT_LIB=tools/lib/libosk_nopk.so timeout 600 python tools/xproc_probe.py --victim gemv --aggressor none --stream-aggressor gemm256p --iters 4000 2>/dev/null | tail -1 | sed 's/^/{"lib": "nopk", "r": /; s/$/}/' >> $O/r06a_xproc_nopk.jsonl
timeout 600 python tools/xproc_probe.py --victim gemv --aggressor gemm256p --iters 4000 2>/dev/null | tail -1 | sed 's/^/{"lib": "shipped", "r": /; s/$/}/' >> $O/r06a_xproc_nopk.jsonl
cat $O/r06a_xproc_nopk.jsonl
timeout 1500 python bench.py --steps 10 --warmup 2 > $O/r06a_bench.json 2> $O/r06a_bench.err
echo "bench rc=$?"; tail -c 3000 $O/r06a_bench.json; tail -5 $O/r06a_bench.err
timeout 1500 bash tools/first_rccl_run.sh 3 1 > $O/r06a_first_rccl_dry.txt 2>&1
echo "dry rc=$?"; cat $O/rccl_first/summary.txt
