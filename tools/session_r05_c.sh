#!/bin/bash
# Round-5 session 3: packed-source timing of the GEMM diagnostics, per-wave stamp rows, micro-batch size around the tile-quantisation points.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python tools/gemm_stamps.py 64 t > gpurun_out/r05_gemm_packed.txt 2>&1 || { echo packed failed; tail -5 gpurun_out/r05_gemm_packed.txt; }
grep -E "timing" gpurun_out/r05_gemm_packed.txt | cut -c1-420
timeout -k 10 200 python tools/gemm_stamps.py 64 > gpurun_out/r05_gemm_stamps_waves.txt 2>&1 || { echo stamps failed; tail -5 gpurun_out/r05_gemm_stamps_waves.txt; }
grep -E "by wave" gpurun_out/r05_gemm_stamps_waves.txt | cut -c1-330 | head -8
for b in 64 65 87 64; do
  timeout -k 10 300 python bench.py --batch $b --steps 30 --warmup 4 --distinct-batches 4 --no-cpu-baseline --no-f16-leg --aligned-utts 0 > gpurun_out/r05_bench_b$b.json 2> gpurun_out/r05_bench_b$b.err || { echo "bench b=$b failed"; tail -3 gpurun_out/r05_bench_b$b.err; }
  python -c "import json; d=json.load(open('gpurun_out/r05_bench_b$b.json')); print('B=$b', d['value'], d['unit'], d['ms_per_step'])"
done
