#!/usr/bin/env bash
# Round-3 counter evidence for the headline kernel (field_fwd_x6_kernel) at the bench's launch shapes, every counter
# group in a pass of its own (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2), plus the kernel-trace statistics of the
# bench command with the bench line of that very run.
#   bash tools/pmc_r03.sh <tag>       (GPU box, repo root)  -> gpurun_out/<tag>_*.txt / .csv / .json
set -uo pipefail
TAG=${1:?tag}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
run() {  # name, counters, bench args...
  local name=$1 ctrs=$2; shift 2
  rm -rf "$OUT/${TAG}_pmc_$name"
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/${TAG}_pmc_$name" -- python3 $REPO/bench.py --steps 1 --warmup 0 \
    --cpu-rays 0 --no-split-legs --no-train-leg "$@" > "$OUT/${TAG}_pmc_$name.log" 2>&1 || { echo "pass $name FAILED"; tail -3 "$OUT/${TAG}_pmc_$name.log"; }
  python3 $REPO/tools/pmc_summary.py "$OUT/${TAG}_pmc_$name" field composite > "$OUT/${TAG}_pmc_$name.txt"
  rm -rf "$OUT/${TAG}_pmc_$name"
  echo "== $name ($ctrs; bench $*)"; cat "$OUT/${TAG}_pmc_$name.txt"
}
run x6_fetch "FETCH_SIZE" --precision bf16x6
run x6_write "WRITE_SIZE" --precision bf16x6
run x6_l2    "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" --precision bf16x6
run x6_sq    "$SQ" --precision bf16x6
run fp32_sq  "$SQ" --precision fp32
# kernel-trace statistics of the default bench command (headline region + the other legs), and its JSON line
rm -rf "$OUT/${TAG}_trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- python3 $REPO/bench.py --steps 3 --warmup 1 \
  --cpu-rays 0 --no-train-leg > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_trace.log" || tail -3 "$OUT/${TAG}_trace.log"
find "$OUT/${TAG}_trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_bench_kernel_stats.csv" \;
rm -rf "$OUT/${TAG}_trace"
head -8 "$OUT/${TAG}_bench_kernel_stats.csv"
