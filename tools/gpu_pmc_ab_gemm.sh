#!/bin/bash
# SQ counters of this library's GEMM and of hipBLASLt at one shape (single-arm runs of tools/ab_vendor.py --one), side by side.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/pmc_ab; rm -rf $O; mkdir -p $O
SHAPE="${GEMM_SHAPE:-8192x8192x8192}"
for arm in osk vendor; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
             "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" \
             "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${arm}_p$i -o p -- python tools/ab_vendor.py --one gemm:$arm:$SHAPE > $O/${arm}_p$i.log 2>&1
  done
done
python - "$O" <<'PY' | tee $O/summary.txt
import csv, glob, collections, sys, os
O = sys.argv[1]
table = collections.defaultdict(dict)
for d in sorted(glob.glob(O + "/*_p*/")):
    arm = os.path.basename(d.rstrip("/")).split("_")[0]
    dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    if not dur: continue
    name = max(dur, key=lambda k: sum(dur[k]))
    table[arm]["kernel"] = name[:70]
    table[arm].setdefault("duration_us", []).append(sum(dur[name][1:]) / max(1, len(dur[name]) - 1) / 1e3)
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"] == name:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        table[arm][k] = sum(v[1:]) / max(1, len(v) - 1)
keys = sorted({k for a in table.values() for k in a if k not in ("kernel", "duration_us")})
for a in table: print(a, table[a]["kernel"], "avg us per pass:", [round(x, 1) for x in table[a]["duration_us"]])
print(f'{"counter":28s} {"osk":>14s} {"vendor":>14s} {"osk/vendor":>10s}')
for k in keys:
    o, v = table.get("osk", {}).get(k), table.get("vendor", {}).get(k)
    if o is None or v is None: continue
    print(f"{k:28s} {o:14.5g} {v:14.5g} {o / v if v else float('nan'):10.3f}")
PY
rm -rf $O/*/
