cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'dit_step', round(d['dit_step_ms'],3), 'gemm_ms', round(d['kernel_cells']['gemm_asm16_kernel [linear]']['ms_total'],2), 'attn', round(d['roofline_self_attention']['avg_launch_ms']*1e3,1))"; }
run off
LTX_WPREFETCH=1 run on
run off2
LTX_WPREFETCH=1 run on2
