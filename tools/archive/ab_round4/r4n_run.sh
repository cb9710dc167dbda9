cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_q2fold.py -x -q -k "attention or cross or fold" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']['attn_cross64_kernel (cross attention)']; print('$1', round(d['value'],1), 'cross_ms', round(k['ms_total'],3), 'avg_us', round(k['avg_ms']*1e3,2))"; }
run default_wps3_g20
LTXHIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libltxhip_xattn_wps2.so run r3_wps2_g13
for g in 13 16 24; do LTX_ATTN_CROSS_GROUPS=$g run wps3_g$g; done
LTX_Q2_FOLD=0 run wps3_g20_nofold
run default_wps3_g20_again
