#!/bin/bash
# Round-5 session 2: the full GPU suite after the constructor-default flip / inexact-weight refusal / switch API, the attention pass ablation with head-score
# deviations (masks 0, 1, 8, 9), and the GEMM wrap-kind / start-spread timings.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 500 python tools/gemm_stamps.py 64 t > gpurun_out/r05_gemm_wrap_kinds.txt 2>&1 || { echo wrap kinds failed; tail -5 gpurun_out/r05_gemm_wrap_kinds.txt; }
grep -E "timing|spread" gpurun_out/r05_gemm_wrap_kinds.txt | cut -c1-330
for legs in "301 10000" "700 10301"; do
  set -- $legs
  timeout -k 10 300 python tools/precision_ablation.py --utts $1 --first-id $2 --steps 8 --attn-drop 0,1,8,9,4,2 --out gpurun_out/r05_attn_pass_scores_$1.txt > gpurun_out/r05_attn_pass_scores_$1.log 2>&1 || { echo ablation $1 failed; tail -5 gpurun_out/r05_attn_pass_scores_$1.log; }
  cut -c1-260 gpurun_out/r05_attn_pass_scores_$1.txt
done
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r05_gpu_tests_b.log 2>&1
echo "gpu suite rc=$?"; tail -15 gpurun_out/r05_gpu_tests_b.log
