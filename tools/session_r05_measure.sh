#!/bin/bash
# Round-5 measurement session (one gpurun call): K-step stamps of the current pair GEMM (real walk vs L2-resident wrap), PMC passes on
# the pair GEMMs and on attn_split32_kernel, and the product-level ablation of the attention's six passes. Outputs under gpurun_out/.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python tools/gemm_stamps.py > gpurun_out/r05_gemm_stamps.txt 2>&1 || { echo stamps failed; tail -5 gpurun_out/r05_gemm_stamps.txt; exit 1; }
cat gpurun_out/r05_gemm_stamps.txt | cut -c1-260
for legs in "301 10000" "700 10301"; do
  set -- $legs
  timeout -k 10 400 python tools/precision_ablation.py --utts $1 --first-id $2 --steps 12 --attn-drop 0,4,8,12,1,2,3,15 --out gpurun_out/r05_attn_pass_ablation_$1.txt > gpurun_out/r05_attn_pass_ablation_$1.log 2>&1 || { echo ablation $1 failed; tail -5 gpurun_out/r05_attn_pass_ablation_$1.log; exit 1; }
  cat gpurun_out/r05_attn_pass_ablation_$1.txt | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
pmc() {  # name, workload..., counters in $PMC
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$ROOT/gpurun_out/pmc_$name" -- python3 "$@" > "$ROOT/gpurun_out/pmc_$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -3 "$ROOT/gpurun_out/pmc_$name.log"; return 0; }
}
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE" pmc gemm_a "$ROOT/tools/gemm_pmc.py"
PMC="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" pmc gemm_b "$ROOT/tools/gemm_pmc.py"
PMC="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum GRBM_GUI_ACTIVE" pmc gemm_c "$ROOT/tools/gemm_pmc.py"
PMC="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE" pmc gemm_d "$ROOT/tools/gemm_pmc.py"
PMC="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" pmc gemm_e "$ROOT/tools/gemm_pmc.py"
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE" pmc attn_a "$ROOT/tools/split_bench.py" 64 1 attn
PMC="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" pmc attn_b "$ROOT/tools/split_bench.py" 64 1 attn
cd "$ROOT"
{
  echo "# rocprofv3 --pmc passes (tools/session_r05_measure.sh) on the contract mode's pair GEMMs (tools/gemm_pmc.py: product kernels, M = 96000; fc1 = <4, true, 1, 0, true>,"
  echo "# qkv = <4, false, 1, 0, true>, fc2 = <2, false, 1, 0, true>); averages per dispatch. SQ_* wave counters in quad-cycles summed over waves; GRBM_GUI_ACTIVE / 8 = kernel cycles."
  for p in a b c d e; do python tools/pmc_summary.py gpurun_out/pmc_gemm_$p "gemm256p_f16_kernel<4, true" "gemm256p_f16_kernel<4, false" "gemm256p_f16_kernel<2, false"; done
} > gpurun_out/r05_gemm_pmc.txt 2>&1
{
  echo "# rocprofv3 --pmc passes on tools/split_bench.py 64 1 attn: the kernel the contract headline runs (attn_split32_kernel<0>), the 16x16x32 pair kernel and the f16 attn32_kernel"
  for p in a b; do python tools/pmc_summary.py gpurun_out/pmc_attn_$p "attn_split32_kernel" "attn_split_kernel" "attn32_kernel"; done
} > gpurun_out/r05_attn_split32_pmc.txt 2>&1
cat gpurun_out/r05_gemm_pmc.txt | head -80
cat gpurun_out/r05_attn_split32_pmc.txt | head -60
rm -rf gpurun_out/pmc_gemm_? gpurun_out/pmc_attn_?
