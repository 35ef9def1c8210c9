#!/usr/bin/env bash
# SQ counters of the training step's kernels (default grade: bf16x6 forward and backward), one pass per counter group.
#   bash tools/pmc_train_sq.sh <tag>     (GPU box, repo root)  -> gpurun_out/<tag>_pmc_train_sq.txt
set -uo pipefail
TAG=${1:?tag}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
: > "$OUT/${TAG}_pmc_train_sq.txt"
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
            "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM"; do
  d="$OUT/${TAG}_pmc_train_sq_tmp"; rm -rf "$d"
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$d" -- python3 $REPO/tools/bench_train.py --steps 2 --instance --penalize \
    > "$d.log" 2>&1 || { echo "pass FAILED ($CTRS)"; tail -3 "$d.log"; }
  python3 $REPO/tools/pmc_summary.py "$d" wgrad field_bwd field_fwd >> "$OUT/${TAG}_pmc_train_sq.txt"
  rm -rf "$d"
done
cat "$OUT/${TAG}_pmc_train_sq.txt"
