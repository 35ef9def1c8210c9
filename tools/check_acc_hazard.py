#!/usr/bin/env python3
"""Static check of the x3 kernels' ISA: every v_accvgpr_read of an accumulator register must be separated from the
last MFMA that wrote its tuple by >= 11 wait states (instructions count 1, s_nop N counts N + 1) -- the inline-asm
reads of field_x3.h are not covered by the compiler's hazard recognizer.
    hipcc ... -S --cuda-device-only -o x.s field_fwd_x3.hip && python tools/check_acc_hazard.py x.s"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
starts = [i for i, l in enumerate(lines) if l.startswith('_Z') and 'x3_kernel' in l]
worst_all = None
for i0 in starts:
    body = []
    for l in lines[i0 + 1:]:
        s = l.strip()
        if s.startswith('.Lfunc_end'):
            break
        if s and not s.startswith((';', '.')) and not s.split()[0].endswith(':'):
            body.append(s)
    last = {}
    worst = 10 ** 9
    reads = 0
    for j, l in enumerate(body):
        m = re.match(r'v_mfma\S* a\[(\d+):(\d+)\]', l)
        if m:
            last[(int(m.group(1)), int(m.group(2)))] = j
            continue
        m = re.match(r'v_accvgpr_read_b32 v\d+, a(\d+)', l)
        if m:
            a = int(m.group(1))
            for (lo, hi), jw in last.items():
                if lo <= a <= hi:
                    reads += 1
                    ws = sum((int(x.split()[1]) + 1) if x.startswith('s_nop') else (8 if x.startswith('v_mfma') else 1)
                             for x in body[jw + 1:j])
                    worst = min(worst, ws)
    if reads == 0:
        continue
    print(f"{lines[i0][:100]}  accumulator reads {reads}, fewest wait states after the producing MFMA: {worst}")
    worst_all = worst if worst_all is None else min(worst_all, worst)
sys.exit(0 if worst_all is not None and worst_all >= 11 else 1)
