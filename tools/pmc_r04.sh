#!/usr/bin/env bash
# Round-4 counter evidence at the bench's launch shapes, every counter group in a pass of its own (TCC has 4 slots:
# FETCH_SIZE takes 3, WRITE_SIZE 2): the headline kernel (field_fwd_x6w_kernel), its 32x32x16 sibling beside it for the SQ
# counters, the training step's weight-gradient traffic in the default (bf16x6) backward, and the kernel-trace statistics
# of the bench command with the bench line of that very run.
#   bash tools/pmc_r04.sh <tag>       (GPU box, repo root)  -> gpurun_out/<tag>_*.txt / .csv / .json
set -uo pipefail
TAG=${1:?tag}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM"
run() {  # name, counters, bench args...
  local name=$1 ctrs=$2; shift 2
  rm -rf "$OUT/${TAG}_pmc_$name"
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/${TAG}_pmc_$name" -- python3 $REPO/bench.py --steps 1 --warmup 0 \
    --cpu-rays 0 --no-split-legs --no-train-leg --no-side-legs "$@" > "$OUT/${TAG}_pmc_$name.log" 2>&1 || { echo "pass $name FAILED"; tail -3 "$OUT/${TAG}_pmc_$name.log"; }
  python3 $REPO/tools/pmc_summary.py "$OUT/${TAG}_pmc_$name" field composite > "$OUT/${TAG}_pmc_$name.txt"
  rm -rf "$OUT/${TAG}_pmc_$name"
  echo "== $name ($ctrs; bench $*)"; cat "$OUT/${TAG}_pmc_$name.txt"
}
run x6w_fetch "FETCH_SIZE" --precision bf16x6
run x6w_write "WRITE_SIZE" --precision bf16x6
run x6w_l2    "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" --precision bf16x6
run x6w_sq    "$SQ" --precision bf16x6
run x6w_sq2   "$SQ2" --precision bf16x6
# the training step (4096 rays, all loss heads, bf16x6 forward and backward): FETCH / WRITE of the weight-gradient kernels
for CTR in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$OUT/${TAG}_pmc_train_$CTR"
  rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$OUT/${TAG}_pmc_train_$CTR" -- python3 $REPO/tools/bench_train.py \
    --steps 2 --instance --penalize > "$OUT/${TAG}_pmc_train_$CTR.log" 2>&1 || { echo "train pass $CTR FAILED"; tail -3 "$OUT/${TAG}_pmc_train_$CTR.log"; }
  python3 $REPO/tools/pmc_summary.py "$OUT/${TAG}_pmc_train_$CTR" wgrad field_bwd field_fwd > "$OUT/${TAG}_pmc_train_$CTR.txt"
  rm -rf "$OUT/${TAG}_pmc_train_$CTR"
  echo "== train $CTR"; cat "$OUT/${TAG}_pmc_train_$CTR.txt"
done
# kernel-trace statistics of the bench's headline region alone (warm-up + timed images: every launch of the headline kernel
# is one of the bench's 20-per-image shapes, so its average is comparable with the HIP-event average of the JSON line)
rm -rf "$OUT/${TAG}_trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- python3 $REPO/bench.py --steps 3 --warmup 1 \
  --cpu-rays 0 --no-train-leg --no-side-legs --no-split-legs > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_trace.log" || tail -3 "$OUT/${TAG}_trace.log"
find "$OUT/${TAG}_trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_bench_kernel_stats.csv" \;
rm -rf "$OUT/${TAG}_trace"
head -8 "$OUT/${TAG}_bench_kernel_stats.csv"
