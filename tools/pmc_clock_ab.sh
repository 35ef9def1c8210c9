#!/usr/bin/env bash
# Cycles AND duration of the bf16x6 field kernel for library variants (is a timing difference cycles or clock?): one pass
# with GRBM_GUI_ACTIVE / SQ cycle counters under --kernel-trace on the probe's launches.
#   bash tools/pmc_clock_ab.sh <tag> lib1.so lib2.so ...    -> gpurun_out/<tag>_clock_ab.txt
set -uo pipefail
TAG=${1:?tag}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
: > "$OUT/${TAG}_clock_ab.txt"
for lib in "$@"; do
  export DMNERF_LIB=$REPO/$lib
  d="$OUT/${TAG}_clock_tmp"; rm -rf "$d"
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$d" -- python3 $REPO/tools/x6_probe.py 2 bf16x6 > "$d.log" 2>&1 || tail -3 "$d.log"
  echo "== $(basename $lib)" >> "$OUT/${TAG}_clock_ab.txt"
  python3 - "$d" >> "$OUT/${TAG}_clock_ab.txt" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "field_fwd" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
ctr = collections.defaultdict(dict)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "field_fwd" in r["Kernel_Name"]:
            ctr[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(ctr, key=int):
    c = ctr[k]; ms = dur.get(k)
    if ms:
        print(f"dispatch {k}: {ms:8.3f} ms  GRBM_GUI_ACTIVE/8 {c['GRBM_GUI_ACTIVE']/8:.4g} cycles -> {c['GRBM_GUI_ACTIVE']/8/ms/1e6:.3f} GHz   MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES']/4/c['SQ_WAVE_CYCLES']*100:.1f} % of wave cycles")
PY
  rm -rf "$d"
done
cat "$OUT/${TAG}_clock_ab.txt"
