#!/usr/bin/env bash
# Round-2 counter evidence, every counter group in a pass of its own (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2 --
# asking for both in one pass is what aborted rocprofv3 in round 1 with "exceeds the capabilities of the hardware").
#   bash tools/pmc_r02.sh            (GPU box, repo root)  -> gpurun_out/r02_pmc_*.txt
set -uo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
run() {  # tag, counters, bench args...
  local tag=$1 ctrs=$2; shift 2
  rm -rf "$OUT/r02_pmc_$tag"
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/r02_pmc_$tag" -- python3 $REPO/bench.py --steps 1 --warmup 0 \
    --cpu-rays 0 --no-split-legs --no-train-leg "$@" > "$OUT/r02_pmc_$tag.log" 2>&1 || { echo "pass $tag FAILED"; tail -3 "$OUT/r02_pmc_$tag.log"; }
  python3 $REPO/tools/pmc_summary.py "$OUT/r02_pmc_$tag" field composite > "$OUT/r02_pmc_$tag.txt"
  echo "== $tag ($ctrs; bench $*)"; cat "$OUT/r02_pmc_$tag.txt"
}
run study_fetch  "FETCH_SIZE"  --precision fp32
run study_write  "WRITE_SIZE"  --precision fp32
run replica_fetch "FETCH_SIZE" --precision bf16x3 --workload replica
run replica_write "WRITE_SIZE" --precision bf16x3 --workload replica
run sq_fp32   "$SQ" --precision fp32
run sq_bf16x3 "$SQ" --precision bf16x3
