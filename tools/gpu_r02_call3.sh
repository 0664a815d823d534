#!/bin/bash
# round-2 call 3: K-step schedule A/B, epilogue cost, vendor kernel resources, step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "== pytest gemm"; timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -x -k "gemm" > $O/pytest_c3.log 2>&1; tail -4 $O/pytest_c3.log
for cfg in "PERSIST=0" "SCHED=0" "SCHED=1"; do
  echo "== epilogue bench OSK_GEMM_$cfg"; env OSK_GEMM_$cfg timeout 200 python tools/gemm_epi_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/epi_$cfg.jsonl
done
for cfg in "SCHED=0" "SCHED=1"; do
  echo "== ab vendor OSK_GEMM_$cfg"; env OSK_GEMM_$cfg timeout 300 python tools/ab_vendor.py --out $O/ab_vendor_$cfg.json 2>&1 | grep -v amdgpu.ids | cut -c1-330
done
echo "== vendor kernel resources"; rm -rf $O/vk; timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/vk -o p -- python tools/ab_vendor.py --one gemm:vendor:8192x8192x8192 > $O/vk.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/vk/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    seen = set()
    for r in rows:
        if r["Kernel_Name"] in seen: continue
        seen.add(r["Kernel_Name"])
        print({k: r[k] for k in r if k in ("Kernel_Name", "Workgroup_Size_X", "Grid_Size_X", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")})
PY
rm -rf $O/vk
for cfg in "OSK_GEMM_PERSIST=0" "OSK_GEMM_SCHED=0" "OSK_GEMM_SCHED=1"; do
  echo "== bench $cfg"; env $cfg timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-b1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
echo "== done"
