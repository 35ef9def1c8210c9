#!/usr/bin/env bash
# Same-device A/B of library variants on the field-kernel probe (tools/x6_probe.py): interleaved rounds in one call.
#   bash tools/ab_x6.sh MODE lib1.so lib2.so ...     (libraries from tools/ab_build.py)
set -uo pipefail
cd "$(dirname "$0")/.."
MODE=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    echo -n "rep$rep $(basename $lib) "
    DMNERF_LIB=$PWD/$lib python tools/x6_probe.py 3 $MODE 2>/dev/null | grep "coarse + fine"
  done
done
