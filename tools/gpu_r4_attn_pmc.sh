#!/bin/bash
# round 4: counters of the wide attention kernel (separate --pmc passes) and per-kernel stats of the step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r04f; rm -rf $O; mkdir -p $O
rm -rf $O/prof_dit
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dit -o dit -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-b1 --no-extra > $O/prof_dit.json 2> $O/prof_dit.err
f=$(find $O/prof_dit -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/step_kernel_stats.csv && head -14 "$f" | cut -c1-170; rm -rf $O/prof_dit
P=$O/pmc_attn; mkdir -p $P; i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $P/p$i -o p$i -- python tools/attn_only.py 3 16 16896 72 > $P/p$i.log 2>&1
done
python - "$P" <<'PY' | tee $O/attn_pmc_summary.txt
import csv, glob, collections, sys, re
for f in sorted(glob.glob(sys.argv[1] + "/p*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            kn = re.search(r"(attn_\w+)", r["Kernel_Name"]).group(1)
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
rm -rf $P/p*/
