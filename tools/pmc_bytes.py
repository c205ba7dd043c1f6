#!/usr/bin/env python3
"""HBM-side bytes and GB/s per launch of the HBM-bound kernels, from the two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; unit
1024 B; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950: 128-byte requests tallied as 64 B; Infinity-Cache hits are counted,
so this is L2 <-> fabric traffic, an upper bound of HBM bytes). Durations from the passes' own dispatch timestamps.
usage: pmc_bytes.py <FETCH_SIZE pass dir> <WRITE_SIZE pass dir> <kernel substring> [...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(lambda: [0.0, 0.0, set()])   # name -> [counter sum, duration sum (s), dispatches]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            key = (f, r["Dispatch_Id"])
            if key not in a[2]:
                a[2].add(key)
                a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print("%-70s %6s %9s %12s %12s %10s" % ("kernel", "calls", "avg us", "fetch MB x2", "write MB", "GB/s"))
for needle in sys.argv[3:]:
    for name in sorted(fetch):
        if needle not in name:
            continue
        f, w = fetch[name], write.get(name, [0.0, 0.0, set()])
        n = max(len(f[2]), 1)
        fb, wb = 2.0 * f[0] * 1024.0 / n, w[0] * 1024.0 / max(len(w[2]), 1)
        dur = f[1] / n
        print("%-70s %6d %9.1f %12.1f %12.1f %10.0f" % (name[:70], n, dur * 1e6, fb / 1e6, wb / 1e6, (fb + wb) / dur / 1e9))
