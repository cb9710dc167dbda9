cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('$1', round(d['value'],1), 'dit_step', round(d['dit_step_ms'],3), 'rownorm_ms', round(k['rownorm_kernel<bf16>']['ms_total'],3), 'gemm_ms', round(d['kernel_cells']['gemm_asm16_kernel [linear]']['ms_total'],2))"; }
run off
for R in 2 4 8; do LTX_NORM_PRESUM=1 LTX_NORM_PRESUM_R=$R run on_R$R; done
run off2
