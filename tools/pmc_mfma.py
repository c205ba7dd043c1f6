#!/usr/bin/env python3
"""MFMA-pipe utilisation and clock per kernel from ONE rocprofv3 --pmc pass with timestamps
(`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace`).
  clock        = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration              (MI355X_MICROARCH.md, DVFS give-back)
  mfma busy    = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)   (the counter counts pipe cycles summed over the SIMDs)
  wait / active = SQ_WAIT_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (quad-cycle units cancel)
usage: pmc_mfma.py <rocprofv3 output dir> [min total ms per kernel]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = defaultdict(dict)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        key = (f, r["Dispatch_Id"])
        rows[key]["name"] = r["Kernel_Name"]
        rows[key]["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        rows[key][r["Counter_Name"]] = rows[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = defaultdict(lambda: defaultdict(float))
for v in rows.values():
    a = agg[v["name"]]
    a["n"] += 1
    for k, x in v.items():
        if k != "name":
            a[k] += x
print("%-86s %6s %9s %8s %9s %7s %7s" % ("kernel", "calls", "avg us", "clk GHz", "mfma busy", "wait", "active"))
for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
    if a["dur"] * 1e3 < min_ms:
        continue
    cyc = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    clk = cyc / a["dur"] / 1e9 if a["dur"] > 0 else 0.0
    busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0) if cyc > 0 else 0.0
    wc = a.get("SQ_WAVE_CYCLES", 0.0)
    print("%-86s %6d %9.1f %8.2f %8.1f%% %7.2f %7.2f" % (name[:86], a["n"], a["dur"] / a["n"] * 1e6, clk, 100 * busy,
                                                         a.get("SQ_WAIT_ANY", 0.0) / wc if wc else 0.0, a.get("SQ_ACTIVE_INST_ANY", 0.0) / wc if wc else 0.0))
