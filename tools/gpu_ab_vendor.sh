#!/bin/bash
# A/B against the vendor kernels (tools/ab_vendor.py) + effective clock of both arms:
# GRBM_GUI_ACTIVE / kernel duration from `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE` passes of single-arm runs.
# Result: gpurun_out/ab_vendor.json (copy to profiles/r02_ab_vendor.json).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/ab_vendor; rm -rf $O; mkdir -p $O
timeout 400 python tools/ab_vendor.py --out gpurun_out/ab_vendor.json > $O/timing.log 2> $O/timing.err; tail -9 $O/timing.log; tail -2 $O/timing.err
SPECS="${AB_SPECS:-gemm:osk:50688x4608x1152 gemm:vendor:50688x4608x1152 gemm:osk:8192x8192x8192 gemm:vendor:8192x8192x8192 attn:osk:3x16x16896x72 attn:vendor:3x16x16896x72}"
for spec in $SPECS; do
  d=$O/$(echo $spec | tr ':' '_')
  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $d -o p -- python tools/ab_vendor.py --one $spec > $d.log 2>&1
done
python - "$O" <<'PY' | tee $O/clocks.txt
import csv, glob, collections, json, os, sys
O = sys.argv[1]
clocks = {}
for d in sorted(glob.glob(O + "/*/")):
    spec = os.path.basename(d.rstrip("/"))
    dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    if not dur:
        continue
    # the arm's dominant kernel = the one with the largest total time
    name = max(dur, key=lambda k: sum(dur[k]))
    cnt = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"] == name and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt.append(float(r["Counter_Value"]))
    n = min(len(cnt), len(dur[name]))
    if n == 0:
        continue
    # skip the first (cold) launch
    c, t = cnt[1:n] or cnt[:n], dur[name][1:n] or dur[name][:n]
    ghz = sum(c) / sum(t) / 8.0   # the counter is summed over the 8 XCDs
    clocks[spec] = {"kernel": name[:100], "launches": n, "avg_us": round(sum(t) / len(t) / 1e3, 1), "effective_clock_ghz": round(ghz, 3)}
    print(spec, clocks[spec])
p = "gpurun_out/ab_vendor.json"
if os.path.exists(p):
    j = json.load(open(p))
    j["effective_clock"] = {"method": "GRBM_GUI_ACTIVE / kernel duration (rocprofv3 --kernel-trace --pmc), profiled single-arm runs", "arms": clocks}
    json.dump(j, open(p, "w"), indent=1)
PY
rm -rf $O/*/
