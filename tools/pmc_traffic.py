#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass) of a bench.py run into
profiles/rNN_dominant_kernel_traffic.json: average bytes per launch of the dominant kernel, FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests tallied as 64 B).
usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [kernel substring]"""
import csv
import glob
import json
import os
import sys


def per_launch(d, counter, needle):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + d)
    per_dispatch = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter or needle not in r.get("Kernel_Name", ""):
                continue
            key = (f, r.get("Dispatch_Id"))
            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(r["Counter_Value"])
    if not per_dispatch:
        raise SystemExit("no %s rows for kernel %r" % (counter, needle))
    v = list(per_dispatch.values())
    return sum(v) / len(v), len(v)


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    needle = sys.argv[4] if len(sys.argv) > 4 else "gemm256p_f16_kernel<0, true, 1, false>"
    fetch_kb, n = per_launch(fetch_dir, "FETCH_SIZE", needle)
    write_kb, _ = per_launch(write_dir, "WRITE_SIZE", needle)
    # rocprofv3 reports both derived counters in KiB-like units of 1024 bytes? No: in KB = 1000? They are "KBytes" = value * 1024 / 1024:
    # FETCH_SIZE = TCC_EA0_RDREQ_32B*32 + (RDREQ - RDREQ_32B)*64 bytes / 1024 -> kilobytes of 1024 bytes.
    fetch_b, write_b = fetch_kb * 1024.0, write_kb * 1024.0
    M, N, K = 96000, 4096, 1024
    res = {
        "kernel": needle, "M": M, "N": N, "K": K, "launches_averaged": n,
        "FETCH_SIZE_raw_bytes": fetch_b, "FETCH_SIZE_corrected_x2_bytes": 2.0 * fetch_b, "WRITE_SIZE_bytes": write_b,
        "traffic_bytes_per_launch": 2.0 * fetch_b + write_b,
        "algorithmic_bytes_per_launch": (M * K + N * K + M * N) * 2 + N * 4,
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 1 --warmup 1 "
                "--no-cpu-baseline` (batch 64); FETCH_SIZE doubled per the gfx950 correction (128-B requests tallied as 64 B). "
                "FETCH counts L2 misses including Infinity-Cache hits, i.e. it is L2<->fabric traffic, an upper bound of HBM bytes.",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
