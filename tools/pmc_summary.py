#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files.

    python tools/pmc_summary.py DIR [substring ...]      # kernels whose name contains any substring (default: all)

Prints `kernel counter per-dispatch-average dispatches`.  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB; the
gfx950 corrections (MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads,
i.e. half the bytes; WRITE_SIZE is exact for 16-B-per-lane streaming stores) are applied by the caller, not here."""
import collections
import csv
import glob
import sys


def summarise(directory, needles=()):
    out = {}
    for path in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(collections.Counter)
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if needles and not any(n in k for n in needles):
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
        for k, d in acc.items():
            for c, v in d.items():
                out[(k, c)] = (v / cnt[k][c], cnt[k][c])
    return out


if __name__ == "__main__":
    res = summarise(sys.argv[1], sys.argv[2:])
    for (k, c), (avg, n) in sorted(res.items()):
        print(f"{k[:70]:70s} {c:28s} {avg:16.6g} per dispatch, {n} dispatches")
