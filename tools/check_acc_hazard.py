#!/usr/bin/env python3
"""Static check of the split-precision kernels' ISA (x3, x6 and the fused kernel's instantiations of both).

1. Every INLINE-ASM v_accvgpr_read of an accumulator register (between ;;#ASMSTART / ;;#ASMEND: the reads of field_x3.h /
   field_x6.h / field_fwd_x6w.hip, which the compiler's hazard recognizer does not cover) must be separated from the last
   MFMA that wrote its tuple by as many wait states as the compiler itself leaves behind that MFMA shape: 11 for the
   16-pass 32x32x16 forms, 8 for the 8-pass 16x16x32 form (instructions count 1, s_nop N counts N + 1, an MFMA in
   between 8).  Reads the compiler generated are its own business and are not held against these numbers.  Rolled loops are walked twice (every backward branch
   is followed once), so a read at the top of a loop body is also held against the MFMAs at its bottom.
2. --lean KERNEL_SUBSTRING (repeatable): kernels whose name contains it must have no scratch
   (.amdhsa_private_segment_fixed_size 0) and at most 32 v_writelane / v_readlane (SGPR parking: the 580-writelane regression of
   experiments/README.md, "Compiler notes").

    hipcc ... -S --cuda-device-only -o x.s field_fwd_x6.hip && python tools/check_acc_hazard.py x.s --lean x6_kernel"""
import re
import sys

KERNEL_TAGS = ("x3_kernel", "x6_kernel", "x6w_kernel")
MAX_LANE_OPS = 32     # a handful of scalars parked around a tile's epilogue is not the regression (580 per pass)


def min_wait_states(mfma: str) -> int:
    return 8 if "16x16x32" in mfma else 11


def kernel_bodies(lines):
    """name -> list of (label or None, instruction text) of every kernel whose name carries one of KERNEL_TAGS."""
    out = {}
    starts = [i for i, l in enumerate(lines) if re.match(r'_Z\S+:', l) and any(t in l for t in KERNEL_TAGS)]
    for i0 in starts:
        body, label = [], None
        for l in lines[i0 + 1:]:
            s = l.strip()
            if s.startswith('.Lfunc_end'):
                break
            if s.startswith(';;#ASMSTART') or s.startswith(';;#ASMEND'):
                body.append((None, s))          # markers: inline-asm reads are told from the compiler's own
                continue
            if not s or s.startswith(';'):
                continue
            if s.split()[0].endswith(':'):
                label = s.split()[0][:-1]
                continue
            if s.startswith('.'):
                continue
            body.append((label, s))
            label = None
        out[lines[i0].split(':')[0]] = body
    return out


def walk(body):
    """Instruction order with every backward branch followed once."""
    where = {lab: j for j, (lab, _) in enumerate(body) if lab}
    taken, order, j = set(), [], 0
    while j < len(body):
        ins = body[j][1]
        order.append(ins)
        m = re.match(r's_(?:c)?branch\S*\s+(\S+)', ins)
        if m and m.group(1) in where and where[m.group(1)] <= j and j not in taken:
            taken.add(j)
            j = where[m.group(1)]
            continue
        j += 1
    return order


def wait_states(ins):
    if ins.startswith(';'):
        return 0
    if ins.startswith('s_nop'):
        return int(ins.split()[1]) + 1
    return 8 if ins.startswith('v_mfma') else 1


def check_reads(order):
    """(inline-asm reads seen, smallest margin = wait states minus what the producing MFMA shape needs)."""
    last, worst, reads, in_asm = {}, 10 ** 9, 0, False
    for j, l in enumerate(order):
        if l.startswith(';;#ASM'):
            in_asm = l.startswith(';;#ASMSTART')
            continue
        m = re.match(r'(v_mfma\S*) a\[(\d+):(\d+)\]', l)
        if m:
            last[(int(m.group(2)), int(m.group(3)))] = (j, m.group(1))
            continue
        m = re.match(r'v_accvgpr_read_b32 v\d+, a(\d+)', l)
        if m and in_asm:
            a = int(m.group(1))
            for (lo, hi), (jw, shape) in last.items():
                if lo <= a <= hi:
                    reads += 1
                    ws = 0
                    for x in order[jw + 1:j]:
                        ws += wait_states(x)
                        if ws >= 4096:          # far enough: do not sum kernel-long stretches for every read
                            break
                    worst = min(worst, ws - min_wait_states(shape))
    return reads, worst


def scratch_sizes(lines):
    out, name = {}, None
    for l in lines:
        s = l.strip()
        if s.startswith('.amdhsa_kernel '):
            name = s.split()[1]
        elif s.startswith('.amdhsa_private_segment_fixed_size') and name:
            out[name] = int(s.split()[1])
    return out


def main(argv):
    path, lean = argv[1], [argv[i + 1] for i, a in enumerate(argv) if a == '--lean']
    lines = open(path).read().split('\n')
    bodies, scratch = kernel_bodies(lines), scratch_sizes(lines)
    worst_all, ok = None, True
    for name, body in bodies.items():
        reads, worst = check_reads(walk(body))
        lanes = sum(1 for _, s in body if s.startswith(('v_writelane', 'v_readlane')))
        if reads:
            print(f"{name[:100]}  accumulator reads {reads}, smallest margin over the producing MFMA's wait states: {worst}; "
                  f"scratch {scratch.get(name, '?')} B, v_writelane/v_readlane {lanes}")
            worst_all = worst if worst_all is None else min(worst_all, worst)
        if any(t in name for t in lean):
            if scratch.get(name) != 0 or lanes > MAX_LANE_OPS:
                print(f"NOT LEAN: {name[:100]}: scratch {scratch.get(name)} B, v_writelane/v_readlane {lanes}")
                ok = False
    return 0 if ok and worst_all is not None and worst_all >= 0 else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv))
