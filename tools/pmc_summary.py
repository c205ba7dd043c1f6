#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per (kernel substring, counter): average per dispatch.
usage: pmc_summary.py <dir> <kernel substring> [<kernel substring> ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
needles = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(set))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        for n in needles:
            if n in kn:
                acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[n][r["Counter_Name"]].add((f, r.get("Dispatch_Id")))
for n in needles:
    print(n)
    for c in sorted(acc[n]):
        k = max(len(cnt[n][c]), 1)
        print("  %-34s %16.1f per dispatch (%d dispatches)" % (c, acc[n][c] / k, k))
