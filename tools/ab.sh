#!/usr/bin/env bash
# Same-box A/B of library variants (devices differ by several % in clock): interleaved rounds in ONE gpurun call.
#   bash tools/ab.sh PREC lib1.so lib2.so ...        (libraries from tools/ab_build.py)
set -euo pipefail
cd "$(dirname "$0")/.."
PREC=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    DMNERF_LIB=$PWD/$lib python bench.py --steps 3 --warmup 1 --cpu-rays 0 --no-split-legs --no-train-leg --precision $PREC 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep$rep', '$lib'.split('/')[-1], round(d['value']), 'rays/s  field ms', round(d['roofline']['avg_launch_ms'],4))"
  done
done
