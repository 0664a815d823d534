#!/bin/bash
# quick CPU->GPU iteration on the generated attention kernel: regenerate, rebuild, test + microbench variant(s)
python tools/gen_attn_asm.py $GEN_ARGS && python -c "from open_sora_amd.build import build_lib; build_lib(force=True)" || exit 1
/usr/local/graft/bin/gpurun --timeout 600 -- "TEST_VARIANTS=\"${TV:--1:1}\" BENCH_VARIANTS=\"${BV:--1:1}\" bash tools/gpu_attn_round.sh" > /tmp/attn_iter.log 2>&1
grep -E "passed|failed|xl.cfg2|status=" /tmp/attn_iter.log | cut -c1-220
