#!/bin/bash
# round-6 end-of-round evidence in one gpurun call -> gpurun_out/final6/   (SKIP_TESTS=1 / SKIP_PMC=1 shorten it)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/final6; rm -rf $O; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -q -m gpu --maxfail=10 --tb=short -p no:cacheprovider -rx > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
fi
echo "== bench (default line)"; timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; tail -2 $O/bench.err
for w in dit vae 11b; do
  echo "== rocprofv3 kernel stats: $w"; rm -rf $O/prof_$w
  args="--workload $w"; [ $w = 11b ] && args="--model 11B"; [ $w = dit ] && args=""
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o $w -- python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-b1 --no-extra > $O/prof_$w.json 2> $O/prof_$w.err
  f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${w}_kernel_stats.csv && head -8 "$f" | cut -c1-160; rm -rf $O/prof_$w
done
echo "== packed GELU A/B inside the step (shipped vs the scalar-GELU build with the round-5 flags)"
for i in 1 2; do
  for lib in shipped scalar_gelu; do
    if [ $lib = scalar_gelu ]; then export OSK_ALT_LIB=tools/lib/libosk_gemm_scalar_gelu.so; else unset OSK_ALT_LIB; fi
    timeout 600 python tools/step_ab.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-b1 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(json.dumps({'lib':'$lib','run':$i,'ms_per_step':r['ms_per_step'],'attn_ms':r['roofline']['avg_launch_ms'],'gemm_ms_per_step':r['roofline_gemm']['block_linear_ms_per_step'],'gemm_frac':r['roofline_gemm']['frac']}))" >> $O/gelu_packed_ab.jsonl
  done
done
unset OSK_ALT_LIB; cat $O/gelu_packed_ab.jsonl
echo "== projection A/B (round-5 GEMM + v_transpose vs osk_gemm_group_bf16)"; timeout 600 python tools/gemm_group_ab.py > $O/gemm_group_ab.jsonl 2>/dev/null; cat $O/gemm_group_ab.jsonl
[ -n "$SKIP_PMC" ] && { echo "== done (PMC skipped)"; exit 0; }
echo "== PMC attention (wide FAST body, tail split + merge), separate passes"
P=$O/pmc_attn; mkdir -p $P; i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $P/p$i -o p$i -- python tools/attn_only.py 3 16 16896 72 > $P/p$i.log 2>&1
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $P/q$i -o q$i -- python tools/attn_only.py 3 24 16896 128 > $P/q$i.log 2>&1
done
python - "$P" <<'PY' | tee $O/attn_pmc_summary.txt
import csv, glob, collections, sys, re
for f in sorted(glob.glob(sys.argv[1] + "/[pq]*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            kn = re.search(r"(attn_\w+)", r["Kernel_Name"]).group(1)
            agg[(kn, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn, k), v in sorted(agg.items()):
        print(f"{kn:22s} {k:32s} per-launch avg {sum(v)/len(v):.6g}  (n={len(v)})")
for f in sorted(glob.glob(sys.argv[1] + "/[pq]1/*kernel_trace.csv")):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            d[re.search(r"(attn_\w+)", r["Kernel_Name"]).group(1)].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for kn, v in d.items():
        print(f"{kn:22s} duration (profiled pass 1) avg {sum(v)/len(v)/1e3:.1f} us  (n={len(v)})")
PY
rm -rf $P/[pq]*/
echo "== PMC fabric traffic of the GEMM / conv launches (tools/gpu_pmc_kernels.sh)"
bash tools/gpu_pmc_kernels.sh > $O/pmc_kernels.log 2>&1; cp gpurun_out/pmc_gemm_conv/summary.txt $O/pmc_gemm_conv_summary.txt 2>/dev/null; tail -30 $O/pmc_gemm_conv_summary.txt
echo "== done"
