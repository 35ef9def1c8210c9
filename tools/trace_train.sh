#!/usr/bin/env bash
# Kernel trace of the training step (all three loss heads, bf16x3 forward): where the device sits idle between kernels.
#   bash tools/trace_train.sh [precision]     -> gpurun_out/trace_train.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; export TMPDIR=/tmp; cd /tmp
PREC=${1:-bf16x3}
rm -rf $OUT/trace_train
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_train -- python3 $REPO/tools/bench_train.py --steps 6 \
  --precision $PREC --instance --penalize > $OUT/trace_train.log 2>&1
python3 - $OUT/trace_train <<'PY' | tee $OUT/trace_train.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))))
# steps are delimited by the coarse saving forward (first big field kernel of a step)
starts = [i for i, r in enumerate(rows) if 'field_fwd' in r[2] and ', true>' in r[2]]
starts = starts[::2]                      # coarse + fine per step
steps = [(starts[k], starts[k + 1]) for k in range(2, len(starts) - 1)]     # skip the first two steps
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:60]
tot = busy = 0
gaps = collections.Counter(); cnt = collections.Counter(); kt = collections.Counter()
for a, b in steps:
    tot += rows[b][0] - rows[a][0]
    for i in range(a, b):
        busy += rows[i][1] - rows[i][0]
        kt[short(rows[i][2])] += rows[i][1] - rows[i][0]
        g = rows[i + 1][0] - rows[i][1]
        if g > 0:
            gaps[short(rows[i + 1][2])] += g
            cnt[short(rows[i + 1][2])] += 1
n = len(steps)
print(f"steps {n}: {tot / n / 1e6:.3f} ms per step, kernels busy {busy / n / 1e6:.3f} ms, idle {(tot - busy) / n / 1e6:.3f} ms, launches per step {sum(b - a for a, b in steps) / n:.0f}")
print("idle time in front of (ms per step, count per step):")
for k, v in gaps.most_common(25):
    print(f"  {v / n / 1e6:7.3f}  {cnt[k] / n:5.1f}  {k}")
print("kernel time (ms per step):")
for k, v in kt.most_common(14):
    print(f"  {v / n / 1e6:7.3f}  {k}")
PY
