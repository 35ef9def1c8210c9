#!/usr/bin/env bash
# Engine clock and package power while the bench runs each precision (rocm-smi polled from a side process).
set -euo pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out/clockwatch; mkdir -p $OUT
for PREC in ${PRECS:-bf16x6 bf16x3 fp32}; do
  ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.1; done ) > $OUT/$PREC.poll &
  POLL=$!
  python bench.py --steps 8 --warmup 1 --cpu-rays 0 --no-split-legs --no-train-leg --precision $PREC > $OUT/$PREC.json 2>/dev/null || true
  kill $POLL || true; wait $POLL 2>/dev/null || true
  echo "== $PREC"; tail -1 $OUT/$PREC.json | cut -c1-200
  grep -oE "sclk clock level: [0-9]+: \([0-9]+Mhz\)|[0-9.]+ ?W|Power \(W\): [0-9.]+" $OUT/$PREC.poll | sort | uniq -c | sort -rn | head -12
done
