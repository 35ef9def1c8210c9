#!/usr/bin/env bash
# `python bench.py --gpus 2` from a bare command line on ONE GPU, both scalings: the script starts its own two ranks
# (dm_nerf_amd/launch.py), gloo stands in for RCCL so that the ranks can share the device, rank 0 prints one JSON line.
# Run from bash on the GPU box (not from a process that has initialised the GPU: that may not start children there).
#   bash tools/check_bench_launch.sh <tag>        -> gpurun_out/<tag>_launch_{weak,strong}.json, rc 0 when both pass
set -uo pipefail
TAG=${1:?tag}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd "$REPO"
unset WORLD_SIZE RANK LOCAL_RANK MASTER_ADDR MASTER_PORT
export DMNERF_DIST_BACKEND=gloo
rc=0
for mode in weak strong; do
  python bench.py --gpus 2 --steps 2 --warmup 1 --cpu-rays 0 --no-split-legs --no-train-leg --no-side-legs --scaling $mode \
    > "$OUT/${TAG}_launch_$mode.json" 2> "$OUT/${TAG}_launch_$mode.err"
  r=$?
  echo "$mode: rc $r, $(grep -c '^{' "$OUT/${TAG}_launch_$mode.json") JSON line(s)"
  python - "$OUT/${TAG}_launch_$mode.json" $mode <<'PY' || rc=1
import json, sys
lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
assert len(lines) == 1, lines
d = json.loads(lines[0])
assert d["n_gpus"] == 2 and d["scaling"] == sys.argv[2] and d["steps"] == 2, d
rays = 640000 * (2 if sys.argv[2] == "weak" else 1) * d["steps"]
assert abs(d["value"] - rays / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
if sys.argv[2] == "strong":
    assert d["sharded_image_bit_equal_to_one_rank_image"] is True
print("  value", round(d["value"]), "rays/s,", d["config"]["parallelism"])
PY
  [ $r -eq 0 ] || rc=1
done
exit $rc
