#!/bin/bash
# Round-5 evidence run on the GPU box (ONE gpurun call, at the final commit): kernel-trace stats and the FETCH_SIZE / WRITE_SIZE passes first (so that the
# bench line of the same call quotes this round's traffic records), then the GPU suite and the bench line. Outputs under gpurun_out/ (copied into profiles/ afterwards).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT" || exit 1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f16-leg --aligned-utts 0 --no-overlap"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r05_stats" -- python3 $B > "$ROOT/gpurun_out/r05_stats.log" 2>&1 || { echo stats failed; tail -5 "$ROOT/gpurun_out/r05_stats.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$ROOT/gpurun_out/r05_pmc_fetch" -- python3 $B > "$ROOT/gpurun_out/r05_pmc_fetch.log" 2>&1 || { echo fetch pass failed; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$ROOT/gpurun_out/r05_pmc_write" -- python3 $B > "$ROOT/gpurun_out/r05_pmc_write.log" 2>&1 || { echo write pass failed; exit 1; }
cd "$ROOT"
python tools/pmc_bytes.py gpurun_out/r05_pmc_fetch gpurun_out/r05_pmc_write head_stats layernorm_pair aggregate_kernel > gpurun_out/r05_hbm_kernels.txt 2>&1
cat gpurun_out/r05_hbm_kernels.txt
for site in attention fc1 fc2 qkv; do WCA_PRECISION=reference python tools/pmc_traffic.py gpurun_out/r05_pmc_fetch gpurun_out/r05_pmc_write gpurun_out/r05_traffic_${site}_reference.json $site > /dev/null 2>&1 || echo "traffic $site failed"; done
python -c "import json; [print(s, json.load(open('gpurun_out/r05_traffic_%s_reference.json' % s)).get('traffic_bytes_per_launch')) for s in ('attention','fc1','fc2','qkv')]"
cp gpurun_out/r05_traffic_*_reference.json profiles/ 2>/dev/null
f=$(find gpurun_out/r05_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r05_bench_kernel_stats_no_overlap_reference.csv && head -12 "$f" | cut -c1-200
rm -rf gpurun_out/r05_stats gpurun_out/r05_pmc_fetch gpurun_out/r05_pmc_write   # keep the merge-back small: drop the raw per-dispatch traces
timeout -k 10 800 python -m pytest tests -q -m gpu > gpurun_out/r05_final_gpu_tests.log 2>&1
echo "gpu suite rc=$?"; tail -3 gpurun_out/r05_final_gpu_tests.log
timeout -k 10 500 python bench.py --stages > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_stage_ms.txt || { echo bench failed; tail -5 gpurun_out/r05_stage_ms.txt; exit 1; }
cut -c1-300 gpurun_out/r05_bench_line.json
