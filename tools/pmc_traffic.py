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


SITES = {  # site -> (kernel-name substring, lambda(M, d) -> (N, K), read-modify-write f32 residual?)
    "qkv": ("gemm256p_f16_kernel<0, false, 1, 0, 0>", lambda d: (3 * d, d), False),
    "out_proj": ("gemm256p_f16_kernel<3, false, 1, 0, 0>", lambda d: (d, d), True),   # round 3: LayerNorm fused (default)
    "fc1": ("gemm256p_f16_kernel<0, true, 1, 0, 0>", lambda d: (4 * d, d), False),
    "fc2": ("gemm256p_f16_kernel<3, false, 4, 0, 0>", lambda d: (d, 4 * d), True),
    "attention": ("attn32_kernel<false>", None, False),
}


# the pair-operand kernels of the reference precision mode (WCA_PRECISION=reference): A rows and f16 outputs are (hi, lo) pairs
SITES_REFERENCE = {
    "qkv": ("gemm256p_f16_kernel<4, false, 1, 0, 2>", lambda d: (3 * d, d), False),
    "out_proj": ("gemm256p_f16_kernel<2, false, 1, 0, 2>", lambda d: (d, d), True),
    "fc1": ("gemm256p_f16_kernel<4, true, 1, 0, 2>", lambda d: (4 * d, d), False),
    "fc2": ("gemm256p_f16_kernel<2, false, 4, 0, 2>", lambda d: (d, 4 * d), True),
    "attention": ("attn_split32_kernel<0>", None, False),
}


def main():
    """usage: [WCA_PRECISION=reference] pmc_traffic.py <FETCH_SIZE pass dir> <WRITE_SIZE pass dir> <out.json> <site> [batch] [model] [d]"""
    fetch_dir, write_dir, out, site = sys.argv[1:5]
    batch = int(sys.argv[5]) if len(sys.argv) > 5 else 64
    model = sys.argv[6] if len(sys.argv) > 6 else "medium"
    d = int(sys.argv[7]) if len(sys.argv) > 7 else 1024
    precision = os.environ.get("WCA_PRECISION", "f16")
    pair = precision != "f16"
    needle, nk, rmw = (SITES_REFERENCE if pair else SITES)[site]
    fetch_kb, n = per_launch(fetch_dir, "FETCH_SIZE", needle)
    write_kb, _ = per_launch(write_dir, "WRITE_SIZE", needle)
    # rocprofv3's derived FETCH_SIZE / WRITE_SIZE are in units of 1024 bytes
    fetch_b, write_b = fetch_kb * 1024.0, write_kb * 1024.0
    M = batch * 1500
    if nk is not None:
        N, K = nk(d)
        if pair:   # A rows [hi | lo], the PLAIN W read once, f16 outputs as pairs; rmw: f32 residual read + write (the LayerNorm is its own launch)
            algo = M * K * 4 + N * K * 2 + N * 4 + (M * N * 8 if rmw else M * N * 4)
        else:
            algo = (M * K + N * K) * 2 + N * 4 + (M * N * 10 if rmw else M * N * 2)   # rmw: f32 residual read + write, + the fused LayerNorm's f16 output
    else:
        N = K = None
        algo = (M * 3 * d + M * d) * (4 if pair else 2)
    import subprocess
    # the GPU box has no .git: the collecting command passes the hash in (WCA_COMMIT=$(git rev-parse --short HEAD) expanded where the repo is)
    head = os.environ.get("WCA_COMMIT") or subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    res = {
        "site": site, "precision": "reference" if pair else "f16", "kernel": needle, "batch": batch, "model": model, "M": M, "N": N, "K": K, "launches_averaged": n, "commit": head,
        "FETCH_SIZE_raw_bytes": fetch_b, "FETCH_SIZE_corrected_x2_bytes": 2.0 * fetch_b, "WRITE_SIZE_bytes": write_b,
        "traffic_bytes_per_launch": 2.0 * fetch_b + write_b,
        "algorithmic_bytes_per_launch": algo,
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 2 --warmup 1 "
                "--no-cpu-baseline`; FETCH_SIZE doubled per the gfx950 correction (128-B requests tallied as 64 B). "
                "FETCH counts L2 misses including Infinity-Cache hits, i.e. it is L2<->fabric traffic, an upper bound of HBM bytes.",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
