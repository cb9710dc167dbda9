#!/bin/bash
# Round 5: small-M ff2 with its K ranges left to the following row norm (C1), depth-to-space convs measured with their own epilogue.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5i; mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests/test_gpu_defer.py tests/test_gpu_gemm_ring.py tests/test_gpu_determinism.py tests/test_gpu_c1.py tests/test_gpu_q2fold.py tests/test_gpu_models.py tests/test_gpu_ops.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -6 $O/pytest.log
J=$O/ab.jsonl; : > $J
run() { tag=$1; shift; env "$@" python3 $R/bench.py --config $CFG --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = {n.split(' (')[0].split('/')[0]: {'ms': round(v['ms_total'], 2), 'avg_us': round(1e3 * v['avg_ms'], 1)} for n, v in d.get('kernels', {}).items()}
print(json.dumps({'arm': '$tag', 'config': '$CFG', 'frames_per_s': round(d['value'], 2), 'ms_per_video': round(d['ms_per_step'], 2), 'dit_step_ms': round(d.get('dit_step_ms', 0), 3), 'vae_decode_ms': round(d.get('vae_decode_ms', 0), 2), 'plans': d.get('gemm_plans'), 'kernels': k}))" >> $J; }
CFG=c1
run default A=1
run "ff2_defer=0" LTX_OPTIONS=ff2_defer=0
run default-again A=1
run "ff2_defer=0 again" LTX_OPTIONS=ff2_defer=0
CFG=c2; run default A=1
cut -c1-420 $J
python3 tools/t5_time.py 2>/dev/null | tail -2
