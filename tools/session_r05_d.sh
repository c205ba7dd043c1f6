#!/bin/bash
# Round-5 session 4: the three-A-slot ring of the pair GEMM (correctness, A/B against the two-slot rings, stamps) and the static-priority experiment of the pair attention.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_split_gpu.py -q -x -k "gemm" > gpurun_out/r05_ring3_tests.log 2>&1
echo "ring3 gemm tests rc=$?"; tail -4 gpurun_out/r05_ring3_tests.log
timeout -k 10 300 python tools/split_bench.py 64 4 > gpurun_out/r05_split_bench_ring3.txt 2>&1 || { echo split_bench failed; tail -5 gpurun_out/r05_split_bench_ring3.txt; }
grep -E "^gemm|^attn|^layernorm" gpurun_out/r05_split_bench_ring3.txt | cut -c1-330
timeout -k 10 200 python tools/gemm_stamps.py 64 s pair:fc1 > gpurun_out/r05_gemm_stamps_ring3.txt 2>&1 || { echo stamps failed; tail -5 gpurun_out/r05_gemm_stamps_ring3.txt; }
grep -vE "amdgpu.ids" gpurun_out/r05_gemm_stamps_ring3.txt | cut -c1-330
WCA_GEMM_RING=1 timeout -k 10 200 python tools/gemm_stamps.py 64 s pair:fc1 > gpurun_out/r05_gemm_stamps_ring2.txt 2>&1
grep -E "by wave|real walk" gpurun_out/r05_gemm_stamps_ring2.txt | cut -c1-330 | head -3
