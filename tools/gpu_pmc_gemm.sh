#!/bin/bash
# SQ / GRBM counters of gemm256_kernel at one shape (bf16 and fp8): clock (GRBM_GUI_ACTIVE / duration) and MFMA-pipe occupancy
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/pmc_gemm; rm -rf $O; mkdir -p $O
SHAPE="${GEMM_SHAPE:-50688 4608 4608}"
for mode in bf16 fp8; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${mode}_p$i -o p$i -- python tools/gemm_only.py $SHAPE $([ $mode = fp8 ] && echo fp8) > $O/${mode}_p$i.log 2>&1
  done
done
python - "$O" <<'PY' | tee $O/summary.txt
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm256" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    run = f.split("/")[-2]
    for k, v in agg.items():
        print(f"{run:8s} {k:28s} per-launch avg {sum(v)/len(v):.6g}  (n={len(v)})")
for f in sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")):
    d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "gemm256" in r["Kernel_Name"]]
    if d:
        print(f"{f.split('/')[-2]:8s} kernel duration avg {sum(d)/len(d)/1e3:.1f} us (n={len(d)})")
PY
rm -rf $O/*/
