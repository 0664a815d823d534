#!/bin/bash
# PMC counter passes over the attention kernel (separate passes: 8 SQ slots each).  Output: gpurun_out/pmc/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/pmc; rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -c . $O/counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_WAVES SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  OSK_ATTN_VARIANT=${V:-0} timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p$i -- python tools/attn_only.py ${ATTN_SHAPE:-3 16 16896 72} > $O/p$i.log 2>&1
  tail -1 $O/p$i.log | cut -c1-200
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc/p*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k:32s} per-launch avg {sum(v)/len(v):.4g}  (n={len(v)})")
PY
