cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/pmc_gemm_fetch; rm -rf $O; mkdir -p $O
for shape in "50688 3456 1152" "50688 4608 1152" "50688 1152 4608" "50688 8064 1152" "50688 1152 5760"; do
  tag=$(echo $shape | tr ' ' 'x')
  for set in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $set | cut -c1-5)
    timeout 100 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${tag}_$n -o p -- python tools/gemm_only.py $shape > $O/${tag}_$n.log 2>&1
  done
done
python - "$O" <<'PY' | tee $O/summary.txt
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm256" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{f.split('/')[-2]:24s} {k:14s} per-launch avg {sum(v)/len(v):.6g}")
PY
rm -rf $O/*/
