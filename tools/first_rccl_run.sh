#!/bin/bash
# First-contact kit for a multi-GPU MI355X node: ONE lease yields everything the sequence-parallel path has
# never produced on hardware -- the RCCL tests, the 1/2/4/8 scaling curve of BASELINE's metric in every exchange mode with the
# exposed-communication accounting, a kernel trace of the 8-rank step, and the tile-parallel VAE decode.  Everything lands under
# gpurun_out/rccl_first/ (copy what should be judged into profiles/).  On a 1-GPU box it runs the world-1 legs only (a dry run of
# the script itself).  Usage: bash tools/first_rccl_run.sh [steps] [warmup]
# Reference bar: /root/reference/README.md:281-288 (SP scaling 96 / 89 / 75 % at 2 / 4 / 8 GPUs).
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
STEPS=${1:-10}; WARM=${2:-3}
OUT=gpurun_out/rccl_first; mkdir -p "$OUT"
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
echo "GPUs visible: $NGPU" | tee "$OUT/summary.txt"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.txt" 2>&1 || { echo "build failed" | tee -a "$OUT/summary.txt"; exit 1; }
port() { python -c "import socket; s = socket.socket(); s.bind(('127.0.0.1', 0)); print(s.getsockname()[1])"; }
run_n() {   # run_n N <script and args>: one rank per GPU over RCCL, as the driver launches bench.py
  local n=$1; shift
  if [ "$n" -eq 1 ]; then python "$@"; else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$(port)" "$@"; fi
}
# 1. the RCCL tests (world 2 / 4 / 8: both exchange modes, bf16 + fp8, 12 forwards back to back bit-identical, exposed-communication accounting)
if [ "$NGPU" -ge 2 ]; then
  timeout 2400 python -m pytest tests/test_gpu_seqpar_nccl.py -x -q > "$OUT/pytest_nccl.txt" 2>&1; echo "pytest nccl rc=$?" | tee -a "$OUT/summary.txt"
  tail -3 "$OUT/pytest_nccl.txt" | tee -a "$OUT/summary.txt"
else echo "1 GPU: RCCL tests skipped (dry run)" | tee -a "$OUT/summary.txt"; fi
# 2. the scaling curve: bench.py --gpus N, every exchange mode, exposed communication per exchange kind in the line's `rccl` object
for mode in auto allgather ulysses; do
  for n in 1 2 4 8; do
    [ "$n" -gt "$NGPU" ] && continue
    [ "$n" -eq 1 ] && [ "$mode" != auto ] && continue
    OSK_SP_MODE=$mode timeout 1200 bash -c "$(declare -f run_n port); run_n $n bench.py --gpus $n --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extra --no-b1" \
      > "$OUT/bench_${mode}_n$n.json" 2> "$OUT/bench_${mode}_n$n.err"
    echo "bench mode=$mode n=$n rc=$? $(python - "$OUT/bench_${mode}_n$n.json" <<'PY'
import json, sys
try:
    r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"ms_per_step={r['ms_per_step']} value={r['value']} exposed={r.get('rccl', {}).get('exposed_comm_ms_per_step_rank0')}")
except Exception as e:
    print("no line:", e)
PY
)" | tee -a "$OUT/summary.txt"
  done
done
# 3. kernel trace of the widest step (all ranks; rank 0's file is the one to read): RCCL kernels beside the MFMA loops
N=$NGPU; [ "$N" -gt 8 ] && N=8
( cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_n$N" -- bash -c "cd $OLDPWD && $(declare -f run_n port); run_n $N bench.py --gpus $N --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-b1" \
  > "$OLDPWD/$OUT/prof_n$N.log" 2>&1 ); echo "rocprofv3 n=$N rc=$?" | tee -a "$OUT/summary.txt"
# 4. tile-parallel VAE: tiled decode of a 33-frame 720p latent, tiles spread over the ranks (bit-identical across world sizes)
for n in 1 2 4 8; do
  [ "$n" -gt "$NGPU" ] && continue
  timeout 900 bash -c "$(declare -f run_n port); run_n $n tools/vae_tile_parallel_bench.py" > "$OUT/vae_tiles_n$n.json" 2> "$OUT/vae_tiles_n$n.err"
  echo "vae tiles n=$n rc=$? $(tail -1 "$OUT/vae_tiles_n$n.json")" | tee -a "$OUT/summary.txt"
done
echo "done: $OUT/summary.txt"
