set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_split_gpu.py -q -x > gpurun_out/v_split.log 2>&1; echo "split tests rc=$?"; tail -2 gpurun_out/v_split.log
timeout -k 10 300 python tools/precision_ablation.py --utts 301 --steps 12 --rows 'full split' --out gpurun_out/v_abl301.txt > gpurun_out/v_abl301.log 2>&1; echo rc=$?; tail -3 gpurun_out/v_abl301.txt
timeout -k 10 400 python tools/precision_ablation.py --first-id 10301 --utts 700 --steps 4 --rows 'full split' --out gpurun_out/v_abl700.txt > gpurun_out/v_abl700.log 2>&1; echo rc=$?; tail -3 gpurun_out/v_abl700.txt
timeout -k 10 200 python tools/parity_ragged.py --leg A --modes reference > gpurun_out/v_ragA.log 2>&1; tail -1 gpurun_out/v_ragA.log
timeout -k 10 200 python tools/parity_ragged.py --leg B --modes reference > gpurun_out/v_ragB.log 2>&1; tail -1 gpurun_out/v_ragB.log
timeout -k 10 300 python -m pytest tests/test_e2e_gpu.py -q -x -k 'contract_mode or ragged' > gpurun_out/v_gate.log 2>&1; tail -2 gpurun_out/v_gate.log
for v in 1 0; do if [ $v = 1 ]; then export WCA_ATTN_SPLIT_VARIANT=1; else unset WCA_ATTN_SPLIT_VARIANT; fi; timeout -k 10 300 python bench.py --steps 60 --warmup 4 --no-cpu-baseline --no-f16-leg --aligned-utts 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant env $v', d['value'], d['ms_per_step'], d['kernels']['attention']['avg_launch_ms'])"; done
