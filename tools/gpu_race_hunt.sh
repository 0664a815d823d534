#!/bin/bash
# One gpurun call: the sequence-parallel repeatability hunt (tools/sp_race_hunt.py) over contention / instrumentation / poison arms.
# (arms T1-T4 of the original hunt switched kernels through OSK_* environment variables; the library no longer reads any)
#   ARMS="A B C" RUNS=40 bash tools/gpu_race_hunt.sh        -> gpurun_out/race/<arm>.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/race; mkdir -p $O
RUNS=${RUNS:-40}
ARMS="${ARMS:-A B C D E F G H I J}"
run() { name=$1; shift; echo "== $name: $*"; timeout ${ARM_TIMEOUT:-420} python tools/sp_race_hunt.py --runs $RUNS --out $O/$name.json "$@" > $O/$name.log 2>&1; echo "rc $?"; tail -c ${TAILC:-1500} $O/$name.log; echo; }
for a in $ARMS; do case $a in
  A) run A_plain ;;
  B) run B_hammer --hammer matmul,copy ;;
  C) run C_fp8_instr --fp8 --instrument ;;
  D) run D_fp8w_instr --fp8-weights-only --instrument ;;
  E) run E_instr_hammer --instrument --hammer matmul,copy ;;
  F) run F_poison_nan --poison nan ;;
  G) run G_poison_rand --poison rand --instrument ;;
  H) run H_stream_hammer --stream-hammer --instrument ;;
  I) run I_ulysses --mode ulysses --instrument --hammer small ;;
  J) run J_w4 --world 4 --geom 1,4,8,8,64 --instrument ;;
  K) run K_hd128 --name hd128_liger_split --geom 3,2,9,7,22 --instrument --hammer small ;;
  L) run L_fp8_plain --fp8 ;;
  T0) run T0_plain ;;
  T5) run T5_checkpoints --checkpoints ;;
  T6) run T6_instrument --instrument ;;
  T7) run T7_depth_1_0 --depth 1,0 --checkpoints ;;
  T8) run T8_depth_0_1 --depth 0,1 --checkpoints ;;
  P1) ARM_TIMEOUT=120 run P1_poison_nan_blocking --poison nan --runs 3 --env HIP_LAUNCH_BLOCKING=1,AMD_SERIALIZE_KERNEL=3 ;;
  *) echo "unknown arm $a" ;;
esac; done
echo "== done"
