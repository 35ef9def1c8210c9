#!/usr/bin/env bash
# SQ counters of the two field kernels (MFMA-busy share of the wave cycles), one pass per precision.
#   bash tools/pmc_sq.sh <tag>      (GPU box, repo root)  -> gpurun_out/<tag>_pmc_sq_{fp32,bf16x3}.txt
set -uo pipefail
TAG=${1:?tag}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
for prec in fp32 bf16x3; do
  rm -rf "$OUT/${TAG}_pmc_sq_$prec"
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$OUT/${TAG}_pmc_sq_$prec" -- python3 $REPO/bench.py --steps 1 --warmup 0 \
    --cpu-rays 0 --no-split-legs --no-train-leg --precision $prec > "$OUT/${TAG}_pmc_sq_$prec.log" 2>&1 || { echo "pass $prec FAILED"; tail -3 "$OUT/${TAG}_pmc_sq_$prec.log"; }
  python3 $REPO/tools/pmc_summary.py "$OUT/${TAG}_pmc_sq_$prec" field > "$OUT/${TAG}_pmc_sq_$prec.txt"
  echo "== $prec"; cat "$OUT/${TAG}_pmc_sq_$prec.txt"
done
