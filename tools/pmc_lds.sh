#!/usr/bin/env bash
set -euo pipefail
TAG=${1:?tag}; PREC=${2:-fp16x3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*|SQ_INSTS_[A-Z_]*|SQ_WAIT[A-Z_]*|SQ_INST_LEVEL[A-Z_]*|TCP_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > "$OUT/counters_list.txt" || true
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS \
  --output-format csv -d "$OUT/pmclds_${TAG}" -- python3 $REPO/bench.py --steps 1 --warmup 0 --cpu-rays 0 --no-fp32-leg --precision $PREC > "$OUT/pmclds_${TAG}.log" 2>&1 || { echo "pmc failed"; tail -5 "$OUT/pmclds_${TAG}.log"; }
python3 - "$OUT/pmclds_${TAG}" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not fs: print('no counters'); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(fs[0])):
    k = r['Kernel_Name'][:48]; acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k, d in acc.items():
    if 'field' in k:
        for c, v in sorted(d.items()): print(c, 'per-dispatch %.5g' % (v / n[k][c]))
PY
