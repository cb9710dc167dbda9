cd $GRAFT_REPO_ROOT
O=gpurun_out/r4o; mkdir -p $O
python -m pytest tests/test_gpu_q2fold.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('$1', round(d['value'],1), 'dit_step', round(d['dit_step_ms'],3), 'rownorm_ms', round(k['rownorm_kernel<bf16>']['ms_total'],3), 'gemm_ms', round(d['kernel_cells']['gemm_asm16_kernel [linear]']['ms_total'],2), 'xattn', round(k['attn_cross64_kernel (cross attention)']['ms_total'],2))"; }
run presum_off
LTX_NORM_PRESUM=1 run presum_on
run presum_off2
LTX_NORM_PRESUM=1 run presum_on2
