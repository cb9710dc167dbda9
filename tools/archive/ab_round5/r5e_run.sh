#!/bin/bash
# Round 5, fifth box: the residual rows of gemm_asm16's 160 x 256 tile requested inside the K loop: bit identity, same-box A/B
# (experiment build, x_gemm_asm16_resid_prefetch=0/1), the shipped library's bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5e; mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_gemm_asm.py tests/test_gpu_tight.py tests/test_gpu_q2fold.py tests/test_gpu_determinism.py tests/test_gpu_c1.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
J=$O/ab.jsonl; : > $J
run() { tag=$1; shift; env "$@" python3 $R/bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = {n.split(' (')[0].split('/')[0]: {'ms': round(v['ms_total'], 2), 'avg_us': round(1e3 * v['avg_ms'], 1)} for n, v in d.get('kernels', {}).items()}
print(json.dumps({'arm': '$tag', 'frames_per_s': round(d['value'], 2), 'ms_per_video': round(d['ms_per_step'], 2), 'dit_step_ms': round(d.get('dit_step_ms', 0), 3), 'vae_decode_ms': round(d.get('vae_decode_ms', 0), 2), 'kernels': k}))" >> $J; }
E=$R/tools/variants/libltxhip_exp.so
run "exp lib, prefetch on" LTXHIP_LIB=$E
run "exp lib, prefetch off" LTXHIP_LIB=$E LTX_OPTIONS=x_gemm_asm16_resid_prefetch=0
run "exp lib, prefetch on (2)" LTXHIP_LIB=$E
run "exp lib, prefetch off (2)" LTXHIP_LIB=$E LTX_OPTIONS=x_gemm_asm16_resid_prefetch=0
run "shipped lib" A=1
cut -c1-330 $J
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-prof > $O/stats.log 2>&1
cd $R
python3 tools/summarize_prof.py $O/stats $O/summary 2>&1 | tail -2
sed -n 1,22p $O/summary.md
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
