#!/bin/bash
# SQ / TCC counters of the conv kernels at single layer shapes (tools/conv_ab.py, CONV_AB_SHAPES) -> gpurun_out/pmc_convsw/summary.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/pmc_convsw; rm -rf $O; mkdir -p $O
export CONV_AB_SHAPES=${CONV_AB_SHAPES:-0,3}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "TCC_HIT_sum TCC_MISS_sum FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  OSK_ALT_LIB=$PMC_LIB timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python tools/conv_ab.py > $O/p$i.log 2>&1
done
python - "$O" <<'PY' | tee $O/summary.txt
import csv, glob, collections, sys, re
O = sys.argv[1]
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(O + "/p*/")):
    disp = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv" in r["Kernel_Name"]:
                k = re.search(r"(conv\w+<[^>]*>)", r["Kernel_Name"]).group(1) + " grid " + r["Grid_Size"] if "Grid_Size" in r else re.search(r"(conv\w+<[^>]*>)", r["Kernel_Name"]).group(1)
                disp[r["Dispatch_Id"]] = k
                tab[k]["dur_us_" + d.rstrip("/")[-2:]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = disp.get(r["Dispatch_Id"])
            if k: tab[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, t in tab.items():
    print("==", k)
    avg = {c: sum(v) / len(v) for c, v in t.items()}
    for c, v in sorted(avg.items()): print("   %-28s %.6g  (n=%d)" % (c, v, len(t[c])))
    if avg.get("GRBM_GUI_ACTIVE") and avg.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        cyc = avg["GRBM_GUI_ACTIVE"] / 8
        print("   -> MFMA pipe busy %.1f %%, clock %.2f GHz" % (100 * avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, cyc / avg["dur_us_p2"] / 1e3))
    if avg.get("SQ_WAVE_CYCLES"):
        print("   -> WAIT_INST_ANY %.1f %% of wave cycles, ACTIVE_INST_ANY %.1f %%" % (100 * avg["SQ_WAIT_INST_ANY"] / avg["SQ_WAVE_CYCLES"], 100 * avg["SQ_ACTIVE_INST_ANY"] / avg["SQ_WAVE_CYCLES"]))
    if avg.get("TCC_HIT_sum"):
        print("   -> TCC hit %.1f %%, fabric-side read %.3f GB (FETCH_SIZE x 2 x 1 KiB), write %.3f GB" % (100 * avg["TCC_HIT_sum"] / (avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"]), avg.get("FETCH_SIZE", 0) * 2 * 1024 / 1e9, avg.get("WRITE_SIZE", 0) * 1024 / 1e9))
PY
rm -rf $O/p*/
