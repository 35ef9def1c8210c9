#!/usr/bin/env bash
# HBM traffic of the training kernels (separate PMC passes; gfx950 FETCH_SIZE counts 32-byte units of wide reads as
# half, see MI355X_MICROARCH.md -> x2 correction applied in the summary below).
set -euo pipefail
TAG=${1:?tag}; PREC=${2:-bf16x3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT/pmctr_${TAG}_$i" -- python3 $REPO/tools/bench_train.py --steps 1 --precision $PREC > "$OUT/pmctr_${TAG}_$i.log" 2>&1 || echo "pass $i failed"
  python3 - "$OUT/pmctr_${TAG}_$i" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not fs: print('no counters'); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(fs[0])):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]; acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k, d in acc.items():
    if 'field' in k or 'wgrad' in k:
        for c, v in sorted(d.items()):
            mb = v / n[k][c] * 1024 / 1e6 * (2 if c == 'FETCH_SIZE' else 1)
            print(f"{k:60s} {c} per dispatch: {mb:10.1f} MB (KB counter x1024{', x2 gfx950 correction' if c == 'FETCH_SIZE' else ''}) dispatches {n[k][c]}")
PY
done
