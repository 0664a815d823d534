#!/bin/bash
# fabric traffic of the fp8 P.V attention kernels at the bench shapes: separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over
# tools/attn_only.py ... pv8 -> gpurun_out/pmc_attn_p8/summary.txt   (records for profiles/attn_traffic.json)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; P=gpurun_out/pmc_attn_p8; rm -rf $P; mkdir -p $P
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $P/p$i -o p$i -- python tools/attn_only.py 3 16 16896 72 pv8 > $P/p$i.log 2>&1
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $P/q$i -o q$i -- python tools/attn_only.py 3 24 16896 128 pv8 > $P/q$i.log 2>&1
done
python - "$P" <<'PY' | tee $P/summary.txt
import csv, glob, collections, sys, re
for f in sorted(glob.glob(sys.argv[1] + "/[pq]*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            kn = re.search(r"(attn_\w+)", r["Kernel_Name"]).group(1)
            agg[(kn, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn, k), v in sorted(agg.items()):
        print(f"{kn:22s} {k:32s} per-launch avg {sum(v)/len(v):.6g}  (n={len(v)})")
PY
rm -rf $P/[pq]*/
