#!/bin/bash
# Round 5: the deferred-ranges row norm with its four ranges unrolled (all loads in flight) - tests, then C1 against the previous build.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5q; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_defer.py tests/test_gpu_c1.py tests/test_gpu_t5.py -q -m gpu -x 2>&1 | grep -E "passed|failed|rror" | tail -3
J=$O/ab.jsonl; : > $J
run() { tag=$1; shift; env "$@" python3 $R/bench.py --config c1 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = {n.split(' (')[0].split('/')[0]: {'ms': round(v['ms_total'], 2), 'avg_us': round(1e3 * v['avg_ms'], 1)} for n, v in d.get('kernels', {}).items()}
print(json.dumps({'arm': '$tag', 'frames_per_s': round(d['value'], 2), 'ms_per_video': round(d['ms_per_step'], 2), 'dit_step_ms': round(d.get('dit_step_ms', 0), 3), 'rownorm': k.get('rownorm_kernel<bf16>'), 'linear': k.get('gemm_asm16_kernel')}))" >> $J; }
run new A=1
run old LTXHIP_LIB=$R/tools/variants/libltxhip_oldgelu.so
run new-again A=1
run old-again LTXHIP_LIB=$R/tools/variants/libltxhip_oldgelu.so
cat $J
python3 tools/t5_time.py 2>/dev/null | tail -1
