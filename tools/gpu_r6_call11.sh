#!/bin/bash
# rocprofv3 evidence for bench.py's rank_shapes (kernel trace + stats of tools/rank_shapes.py) and MFMA-pipe counters of the projection kernels in the step
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
rm -rf $O/prof_rank; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_rank -o rank -- python $OLDPWD/tools/rank_shapes.py > $OLDPWD/$O/r06k_rank_shapes.jsonl 2> $OLDPWD/$O/r06k_rank_shapes.err )
f=$(find $O/prof_rank -name "*kernel_stats.csv" | head -1); cp "$f" $O/r06k_rank_shapes_kernel_stats.csv; head -12 "$f" | cut -c1-200; rm -rf $O/prof_rank
cut -c1-400 $O/r06k_rank_shapes.jsonl
P=$O/pmc_step; rm -rf $P; mkdir -p $P
for i in 1 2; do
  set="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY"; [ $i = 2 ] && set="GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU"
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $P/p$i -o p$i -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-b1 --no-extra > $P/p$i.log 2>&1
done
python - "$P" <<'PY' | tee $O/r06k_pmc_step_mfma.txt
import csv, glob, collections, sys, re
for f in sorted(glob.glob(sys.argv[1] + "/p*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(gemm256x_vt_kernel|gemm256x_kernel<[^>]*>|attn_asm72w_kernel|qknorm_rope_rows_kernel|ln_modulate_kernel)", r["Kernel_Name"])
        if m:
            agg[(m.group(1), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn, k), v in sorted(agg.items()):
        print(f"{kn:40s} {k:28s} per-launch avg {sum(v)/len(v):.6g}  (n={len(v)})")
for f in sorted(glob.glob(sys.argv[1] + "/p1/*kernel_trace.csv")):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(gemm256x_vt_kernel|gemm256x_kernel<[^>]*>|attn_asm72w_kernel)", r["Kernel_Name"])
        if m: d[m.group(1)].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for kn, v in d.items(): print(f"{kn:40s} duration (profiled pass 1) avg {sum(v)/len(v)/1e3:.1f} us (n={len(v)})")
PY
rm -rf $P
