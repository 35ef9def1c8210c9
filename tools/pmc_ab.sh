#!/usr/bin/env bash
# HBM-side fetch bytes and L2 hit rate of the bf16x6 field kernel for several library variants (tools/ab_build.py) on the
# probe's two launch shapes (tools/x6_probe.py: 65536 rays, S = 64 and 192), one counter group per pass.
#   bash tools/pmc_ab.sh <tag> lib1.so lib2.so ...      (GPU box, repo root)  -> gpurun_out/<tag>_pmc_ab.txt
set -uo pipefail
TAG=${1:?tag}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
: > "$OUT/${TAG}_pmc_ab.txt"
for lib in "$@"; do
  export DMNERF_LIB=$REPO/$lib
  for CTRS in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    d="$OUT/${TAG}_pmc_ab_tmp"; rm -rf "$d"
    rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$d" -- python3 $REPO/tools/x6_probe.py 1 bf16x6 \
      > "$d.log" 2>&1 || { echo "pass FAILED ($lib, $CTRS)"; tail -3 "$d.log"; }
    echo "== $(basename $lib)" >> "$OUT/${TAG}_pmc_ab.txt"
    python3 $REPO/tools/pmc_summary.py "$d" field_fwd >> "$OUT/${TAG}_pmc_ab.txt"
    rm -rf "$d"
  done
done
cat "$OUT/${TAG}_pmc_ab.txt"
