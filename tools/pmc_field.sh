#!/usr/bin/env bash
# SQ counter pass over the bench (1 step) for the field kernel; run on the GPU box from the repo root.
set -euo pipefail
TAG=${1:?tag}; PREC=${2:-fp16x3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/pmc_${TAG}" -- python3 $REPO/bench.py --steps 1 --warmup 0 --cpu-rays 0 --no-fp32-leg --precision $PREC > "$OUT/pmc_${TAG}.log" 2>&1
python3 - "$OUT/pmc_${TAG}" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:60]; acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
for k, d in acc.items():
    if 'field' in k:
        print(k, 'dispatches', n[k]); [print('   ', c, '%.4g' % (v / max(n[k], 1))) for c, v in sorted(d.items())]
PY
