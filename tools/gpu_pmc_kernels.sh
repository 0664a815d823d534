#!/bin/bash
# HBM traffic (PMC FETCH_SIZE / WRITE_SIZE, separate passes) of the GEMM and conv kernels: per-kernel-name totals over one
# bench step.  Output: gpurun_out/pmc_gemm_conv/summary.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/pmc_gemm_conv; rm -rf $O; mkdir -p $O
run() {  # name, counters, command...
  local name=$1 set=$2; shift 2
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$name -o $name -- "$@" > $O/$name.log 2>&1
  tail -1 $O/$name.log | cut -c1-160
}
export OSK_BENCH_NO_GN_FOLD=1
if [ -z "$PMC_ONLY_DIT" ]; then
run vae_fetch "FETCH_SIZE" python bench.py --workload vae --steps 1 --warmup 0 --no-cpu-baseline
run vae_write "WRITE_SIZE" python bench.py --workload vae --steps 1 --warmup 0 --no-cpu-baseline
# the same step with the opt-in GroupNorm fold (hunyuan_vae.FOLD_GN)
OSK_VAE_FOLD_GN=1 run vaefold_fetch "FETCH_SIZE" python bench.py --workload vae --steps 1 --warmup 0 --no-cpu-baseline
OSK_VAE_FOLD_GN=1 run vaefold_write "WRITE_SIZE" python bench.py --workload vae --steps 1 --warmup 0 --no-cpu-baseline
if [ -n "$PMC_LINEAR_CONV" ]; then   # the round-1 tile mapping of conv256t for comparison
  run vaelin_fetch "FETCH_SIZE" python bench.py --workload vae --steps 1 --warmup 0 --no-cpu-baseline
  run vaelin_write "WRITE_SIZE" python bench.py --workload vae --steps 1 --warmup 0 --no-cpu-baseline
fi
fi
if [ -z "$PMC_ONLY_VAE" ]; then
run dit_fetch "FETCH_SIZE" python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-b1 --no-extra
run dit_write "WRITE_SIZE" python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-b1 --no-extra
fi
python - "$O" <<'PY' | tee $O/summary.txt
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = next((k for k in ("gemm256x_vt_kernel", "conv_fewout_kernel", "rownorm2_max", "attn_asm72w_kernel", "attn_merge", "convsw2_kernel", "convsw_kernel", "conv256x_kernel", "conv256w_kernel", "conv256t_kernel", "gemm256x_kernel", "conv256_kernel", "conv3d_kernel", "gemm256w_kernel", "gemm256p_kernel",
                                "gemm256_kernel", "gemm_bf16_kernel", "attn_asm72_kernel", "attn_hd512_kernel", "gn_stats",
                                "gn_apply", "qknorm_rope", "ln_modulate", "v_transpose") if k in n), None)
        if key:
            agg[(key, r["Counter_Name"])][0] += float(r["Counter_Value"])
            agg[(key, r["Counter_Name"])][1] += 1
    run = f.split("/")[-2]
    tot = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
    for c, v in sorted(tot.items()):
        print(f"{run:10s} {'ALL KERNELS':20s} {c:12s} total {v:.6g} KB")
    for (k, c), (v, n) in sorted(agg.items()):
        print(f"{run:10s} {k:20s} {c:12s} total {v:.6g} KB over {n} launches")
PY
rm -rf $O/*/
