#!/usr/bin/env bash
# L2 / fabric counters for the field kernel (separate passes: TCC slots are scarce).
set -euo pipefail
TAG=${1:?tag}; PREC=${2:-fp16x3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
i=0
for CTRS in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT/pmcl2_${TAG}_$i" -- python3 $REPO/bench.py --steps 1 --warmup 0 --cpu-rays 0 --no-fp32-leg --precision $PREC > "$OUT/pmcl2_${TAG}_$i.log" 2>&1 || echo "pass $i failed"
  python3 - "$OUT/pmcl2_${TAG}_$i" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not fs: print('no counters'); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(fs[0])):
    k = r['Kernel_Name'][:48]; acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k, d in acc.items():
    if 'field' in k or 'composite' in k:
        for c, v in sorted(d.items()): print(k, c, 'per-dispatch %.5g' % (v / n[k][c]), 'dispatches', n[k][c])
PY
done
