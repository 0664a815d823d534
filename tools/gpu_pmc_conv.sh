#!/bin/bash
# SQ counters of the VAE's conv kernels over one encode + decode: MFMA-pipe busy fraction and clock per kernel name.
# Output: gpurun_out/pmc_conv/summary.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/pmc_conv; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python bench.py --workload vae --steps 1 --warmup 0 --no-cpu-baseline > $O/p$i.log 2>&1
  tail -1 $O/p$i.log | cut -c1-120
done
python - "$O" <<'PY' | tee $O/summary.txt
import csv, glob, collections, sys
O = sys.argv[1]
KEYS = ("conv256w_kernel", "conv256t_kernel<256>", "conv256t_kernel<128>", "conv3d_kernel", "attn_hd512", "gn_stats", "gn_apply")
key_of = lambda n: next((k for k in KEYS if k in n), None)
tab = collections.defaultdict(lambda: collections.defaultdict(float))
for d in sorted(glob.glob(O + "/p*/")):
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = key_of(r["Kernel_Name"])
            if k:
                tab[k]["dur_ns_" + d.rstrip("/")[-2:]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                tab[k]["launches_" + d.rstrip("/")[-2:]] += 1
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = key_of(r["Kernel_Name"])
            if k: tab[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, t in tab.items():
    print("==", k)
    for c, v in sorted(t.items()): print("   %-28s %.6g" % (c, v))
    if t.get("GRBM_GUI_ACTIVE") and t.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        cyc = t["GRBM_GUI_ACTIVE"] / 8      # summed over the 8 XCDs
        print("   -> MFMA pipe busy %.1f %%, clock %.2f GHz (pass-2 duration)" % (100 * t["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, cyc / t["dur_ns_p2"]))
    if t.get("SQ_WAVE_CYCLES"):
        print("   -> waits: WAIT_INST_ANY %.1f %% of wave cycles" % (100 * t["SQ_WAIT_INST_ANY"] / t["SQ_WAVE_CYCLES"]))
    if t.get("TCC_REQ_sum"):
        print("   -> TCC hit %.1f %%" % (100 * t["TCC_HIT_sum"] / max(1.0, t["TCC_HIT_sum"] + t["TCC_MISS_sum"])))
PY
rm -rf $O/p*/
