#!/bin/bash
# Round 5: the generated GEMM loop in conv mode (VAE mid block, conv_in, first upsampler): bit identity, decode tests, bench A/B.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5h; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_gemm_asm.py tests/test_gpu_determinism.py tests/test_gpu_c2.py tests/test_gpu_c1.py tests/test_gpu_tight.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -6 $O/pytest.log
J=$O/ab.jsonl; : > $J
run() { tag=$1; shift; env "$@" python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = {n.split(' (')[0].split('/')[0]: {'ms': round(v['ms_total'], 2), 'avg_us': round(1e3 * v['avg_ms'], 1)} for n, v in d.get('kernels', {}).items()}
print(json.dumps({'arm': '$tag', 'config': '$CFG', 'frames_per_s': round(d['value'], 2), 'ms_per_video': round(d['ms_per_step'], 2), 'dit_step_ms': round(d.get('dit_step_ms', 0), 3), 'vae_decode_ms': round(d.get('vae_decode_ms', 0), 2), 'mid_plan': d.get('gemm_plans', {}).get('vae_mid_1024'), 'cells': {n: round(v['ms_total'], 2) for n, v in d.get('kernel_cells', {}).items() if 'conv' in n}, 'kernels': k}))" >> $J; }
CFG=c2
run default A=1
run "gemm_off=asm16 (mid block on gemm_big)" LTX_OPTIONS=gemm_off=asm16
run default-again A=1
CFG=c5; run default A=1
cut -c1-420 $J
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-prof > $O/stats.log 2>&1
cd $R && python3 tools/summarize_prof.py $O/stats $O/summary 2>&1 | tail -2; grep -n "VAE decode, launch by launch" -A18 $O/summary.md
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
