#!/bin/bash
# The library built WITHOUT the -packed-fp32-ops workaround of open_sora_amd/build.py (hipcc may then emit v_pk_{fma,mul,add}_f32 /
# v_pk_mov_b32 in the compiler-scheduled kernels) -> tools/lib/libosk_nopk.so.  Round 6 re-probe of profiles/r03_cross_kernel_interference.md
# on today's tree (round 5 changed how every asm loop hands its registers to the compiler):
#   OSK_ALT_LIB=tools/lib/libosk_nopk.so python tools/xproc_probe.py --victim gemv --aggressor gemm256p --iters 4000
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/lib /tmp/osk_nopk_obj
pids=()
for f in open_sora_amd/csrc/*.hip; do
  o=/tmp/osk_nopk_obj/$(basename "${f%.hip}").o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o "$o" 2>/dev/null &
  pids+=($!)
  if [ ${#pids[@]} -ge 8 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lib/libosk_nopk.so /tmp/osk_nopk_obj/*.o
echo "built tools/lib/libosk_nopk.so"
/opt/rocm/lib/llvm/bin/llvm-objdump -d --offloading tools/lib/libosk_nopk.so >/dev/null 2>&1 || true
