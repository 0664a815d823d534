#!/bin/bash
# Where the wave cycles of the attention and GEMM loops go: SQ issue / wait counters in separate --pmc passes over tools/attn_only.py and
# tools/gemm_only.py -> gpurun_out/pmc_issue/summary.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; P=gpurun_out/pmc_issue; rm -rf $P; mkdir -p $P
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
           "SQ_WAVE_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_TRANS_F32 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $P/a$i -o a$i -- python tools/attn_only.py 3 16 16896 72 > $P/a$i.log 2>&1
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $P/g$i -o g$i -- python tools/gemm_only.py 50688 8064 1152 > $P/g$i.log 2>&1
done
python - "$P" <<'PY' | tee $P/summary.txt
import csv, glob, collections, sys, re
for f in sorted(glob.glob(sys.argv[1] + "/[ag]*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(attn_asm\w+|gemm256x_kernel)", r["Kernel_Name"])
        if m:
            agg[(m.group(1), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn, k), v in sorted(agg.items()):
        print(f"{kn:22s} {k:32s} per-launch avg {sum(v)/len(v):.6g}  (n={len(v)})")
PY
rm -rf $P/[ag]*/
