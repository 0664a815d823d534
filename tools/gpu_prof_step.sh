#!/bin/bash
# rocprofv3 kernel stats of the bench step (default config); PROF_TAG names the output; extra env passes through
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
TAG=${PROF_TAG:-step}
rm -rf $O/prof_$TAG
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o $TAG -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-b1 ${BENCH_ARGS:-} > $O/prof_$TAG.json 2> $O/prof_$TAG.err
cat $O/prof_$TAG.json | cut -c1-300
f=$(find $O/prof_$TAG -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  {float(r["Percentage"]):6.2f} %')
PY
cp "$f" $O/${TAG}_kernel_stats.csv; rm -rf $O/prof_$TAG
