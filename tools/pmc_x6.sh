#!/usr/bin/env bash
# SQ and L2 counters of one field kernel on the probe's two launch shapes (tools/x6_probe.py), one pass per counter group.
#   bash tools/pmc_x6.sh <tag> [mode]      (GPU box, repo root)  -> gpurun_out/<tag>_pmc_<mode>.txt
set -uo pipefail
TAG=${1:?tag}; MODE=${2:-bf16x6}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
: > "$OUT/${TAG}_pmc_$MODE.txt"
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
            "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf "$OUT/${TAG}_pmc_${MODE}_$i"
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT/${TAG}_pmc_${MODE}_$i" -- python3 $REPO/tools/x6_probe.py 1 $MODE \
    > "$OUT/${TAG}_pmc_${MODE}_$i.log" 2>&1 || { echo "pass $i FAILED"; tail -3 "$OUT/${TAG}_pmc_${MODE}_$i.log"; }
  python3 $REPO/tools/pmc_summary.py "$OUT/${TAG}_pmc_${MODE}_$i" field >> "$OUT/${TAG}_pmc_$MODE.txt"
  rm -rf "$OUT/${TAG}_pmc_${MODE}_$i"
done
cat "$OUT/${TAG}_pmc_$MODE.txt"
