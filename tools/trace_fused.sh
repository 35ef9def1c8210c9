#!/usr/bin/env bash
# kernel-trace of one bf16x3 image, fused and chained, to compare per-launch durations (coarse S=64 vs fine S=192)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; export TMPDIR=/tmp; cd /tmp
for mode in fused nofused; do
  flag="--fused on"; [ $mode = nofused ] && flag="--fused off"
  rm -rf $OUT/r02_trace_$mode
  rocprofv3 --kernel-trace --output-format csv -d $OUT/r02_trace_$mode -- python3 $REPO/bench.py --steps 2 --warmup 1 --cpu-rays 0 --no-split-legs --no-train-leg --precision bf16x3 $flag > $OUT/r02_trace_$mode.log 2>&1
  python3 - $OUT/r02_trace_$mode <<'PY'
import csv, glob, sys, collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'field_f' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in rows]
d=d[len(d)//3:]   # drop warm-up image
small=sorted(x for x in d if x < 15); big=sorted(x for x in d if x >= 15)
med=lambda v: v[len(v)//2] if v else 0
print(sys.argv[1].split('_')[-1], 'launches', len(d), 'coarse-like median ms', med(small), 'n', len(small), '| fine-like median ms', med(big), 'n', len(big), '| sum per image', sum(d)/2)
PY
done
