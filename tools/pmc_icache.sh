#!/usr/bin/env bash
# Instruction-fetch counters for the (fully unrolled, ~150 KB) field kernels.
set -euo pipefail
TAG=${1:?tag}; PREC=${2:-fp16x3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
i=0
for CTRS in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT/pmcic_${TAG}_$i" -- python3 $REPO/bench.py --steps 1 --warmup 0 --cpu-rays 0 --no-fp32-leg --precision $PREC > "$OUT/pmcic_${TAG}_$i.log" 2>&1 || echo "pass $i failed"
  python3 - "$OUT/pmcic_${TAG}_$i" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not fs: print('no counters'); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(fs[0])):
    k = r['Kernel_Name'][:48]; acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k, d in acc.items():
    if 'field' in k:
        for c, v in sorted(d.items()): print(k, c, 'per-dispatch %.5g' % (v / n[k][c]), 'dispatches', n[k][c])
PY
done
