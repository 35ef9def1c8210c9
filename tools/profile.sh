#!/usr/bin/env bash
# Profile the headline bench on the GPU box.  Usage (from the repo root, on the box):
#   bash tools/profile.sh <tag>            # kernel trace + stats  -> gpurun_out/prof_<tag>_trace/
#   bash tools/profile.sh <tag> pmc        # + two PMC passes      -> gpurun_out/prof_<tag>_pmc{1,2}/
# Counters are collected in their own runs (never together with trace domains other than
# --kernel-trace).  Summaries worth keeping are copied into profiles/ by hand and committed.
set -euo pipefail
TAG=${1:?tag}
MODE=${2:-trace}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="$REPO/bench.py --steps 1 --warmup 1 --cpu-rays 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_trace" -- python3 $BENCH \
  > "$OUT/prof_${TAG}_trace.log" 2>&1
if [ "$MODE" = "pmc" ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_${TAG}_pmc1" -- python3 $BENCH \
    > "$OUT/prof_${TAG}_pmc1.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/prof_${TAG}_pmc2" -- python3 $BENCH > "$OUT/prof_${TAG}_pmc2.log" 2>&1
fi
find "$OUT" -name "*stats*.csv" | head
