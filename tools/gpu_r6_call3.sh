#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" > $O/r06c_pytest_gemm.txt 2>&1; echo "pytest gemm rc=$?"; tail -4 $O/r06c_pytest_gemm.txt
timeout 600 python tools/gemm_group_ab.py > $O/r06c_gemm_group_ab.jsonl 2>&1; cat $O/r06c_gemm_group_ab.jsonl
OSK_ALT_LIB=tools/lib/libosk_gemm_timing.so timeout 600 python tools/gemm_tile_timing.py > $O/r06c_gemm_tile_timing.jsonl 2>&1; cut -c1-420 $O/r06c_gemm_tile_timing.jsonl
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $O/r06c_bench.json 2>$O/r06c_bench.err; python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r06c_bench.json') if l.startswith('{')][-1])
print(r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline_gemm']['block_linear_ms_per_step'], r['roofline_gemm']['frac'], r['b1']['ms_per_step'])
PY
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pkfp32_repro.hip -o /tmp/pkfp32_repro -ldl 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops tools/pkfp32_repro.hip -o /tmp/pkfp32_repro_nopk -ldl 2>/dev/null
for arm in "/tmp/pkfp32_repro open_sora_amd/lib/libosk_hip.so" "/tmp/pkfp32_repro_nopk open_sora_amd/lib/libosk_hip.so" "/tmp/pkfp32_repro none"; do
  echo "{\"cmd\": \"$arm 4000\", \"result\": $(timeout 300 $arm 4000 | tail -1)}" >> $O/r06c_pkfp32_repro.jsonl
done
cat $O/r06c_pkfp32_repro.jsonl
timeout 900 python -m pytest tests/test_gpu_vae.py -x -q -m gpu -k "conv or full_size or golden" > $O/r06c_pytest_vae.txt 2>&1; echo "pytest vae rc=$?"; tail -4 $O/r06c_pytest_vae.txt
timeout 600 python bench.py --workload vae --steps 10 --warmup 3 --no-cpu-baseline > $O/r06c_bench_vae.json 2>/dev/null; cut -c1-330 $O/r06c_bench_vae.json
