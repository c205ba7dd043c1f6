#!/bin/bash
# bench A/B of the pair GEMM's LDS ring (form B vs round 4's two-slot rings) in the driver-shaped loop, same box, back to back
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT" || exit 1
mkdir -p gpurun_out
for ring in 0 1 0 1; do
  WCA_GEMM_RING=$ring timeout -k 10 300 python bench.py --steps 40 --warmup 5 --distinct-batches 4 --no-cpu-baseline --no-f16-leg --aligned-utts 0 > gpurun_out/r05_bench_ring$ring.json 2> gpurun_out/r05_bench_ring$ring.err || { echo "bench ring=$ring failed"; tail -3 gpurun_out/r05_bench_ring$ring.err; }
  python -c "import json; d=json.load(open('gpurun_out/r05_bench_ring$ring.json')); print('gemm_ring=$ring', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms/step; parity', d.get('parity',{}).get('boundaries_identical', d.get('parity')))" | cut -c1-300
done
