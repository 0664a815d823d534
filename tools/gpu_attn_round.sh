#!/bin/bash
# One gpurun call focused on the attention kernels: parity tests per variant, microbench per (variant, map),
# PMC passes (SQ + TCC/HBM traffic, separate passes) for $PMC_VARIANTS.  Output under gpurun_out/.
#   TEST_VARIANTS="3:0 4:1" BENCH_VARIANTS="0:0 0:1 3:1" PMC_VARIANTS="0:0" bash tools/gpu_attn_round.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for vm in ${TEST_VARIANTS:-}; do v=${vm%%:*}; m=${vm##*:}
  echo "== pytest attention variant $v map $m"
  OSK_ATTN_VARIANT=$v OSK_ATTN_MAP=$m timeout 400 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k attention --tb=short -p no:cacheprovider > $O/pytest_attn_v${v}m${m}.log 2>&1
  tail -15 $O/pytest_attn_v${v}m${m}.log | cut -c1-300
done
: > $O/microbench_attn.jsonl
for vm in ${BENCH_VARIANTS:-}; do v=${vm%%:*}; m=${vm##*:}
  echo "== microbench attn variant $v map $m"
  OSK_ATTN_VARIANT=$v OSK_ATTN_MAP=$m timeout 200 python tools/microbench.py --quick --attn-only 2>> $O/microbench.err | grep attention_fwd | sed "s/^{/{\"attn_variant\": $v, \"attn_map\": $m, /" | tee -a $O/microbench_attn.jsonl
done
for vm in ${PMC_VARIANTS:-}; do v=${vm%%:*}; m=${vm##*:}
  P=$O/pmc_v${v}m${m}; rm -rf $P; mkdir -p $P
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY" \
             "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum"; do
    i=$((i+1))
    OSK_ATTN_VARIANT=$v OSK_ATTN_MAP=$m timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $P/p$i -o p$i -- python tools/attn_only.py ${ATTN_SHAPE:-3 16 16896 72} > $P/p$i.log 2>&1
    tail -1 $P/p$i.log | cut -c1-200
  done
  python - "$P" <<'PY' | tee $P/summary.txt
import csv, glob, collections, sys
import re
for f in sorted(glob.glob(sys.argv[1] + "/p*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            kn = re.search(r"(attn_\w+)", r["Kernel_Name"]).group(1)      # attn_asm72_kernel / attn_merge_kernel separately
            agg[(kn, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn, k), v in sorted(agg.items()):
        print(f"{kn:22s} {k:32s} per-launch avg {sum(v)/len(v):.6g}  (n={len(v)})")
for f in sorted(glob.glob(sys.argv[1] + "/p1/*kernel_trace.csv")):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            d[re.search(r"(attn_\w+)", r["Kernel_Name"]).group(1)].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for kn, v in d.items():
        print(f"{kn:22s} duration (profiled pass 1) avg {sum(v)/len(v)/1e3:.1f} us  (n={len(v)})")
PY
  rm -rf $P/p*/  # raw per-dispatch csvs are large; the summary is what is kept
done
echo "== done"
