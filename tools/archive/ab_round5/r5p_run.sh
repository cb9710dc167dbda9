#!/bin/bash
# Round 5: GELU epilogue with folded constants on packed f32 operations - tests, then same-box A/B against the previous build (tools/variants/libltxhip_oldgelu.so).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5p; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_gemm_asm.py tests/test_gpu_ops.py tests/test_gpu_tight.py tests/test_gpu_gemm_ring.py tests/test_gpu_models.py -q -m gpu -x 2>&1 | tail -3
J=$O/ab.jsonl; : > $J
run() { tag=$1; shift; env "$@" python3 $R/bench.py --config $CFG --steps 6 --warmup 1 --no-cpu-baseline --no-batched 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = {n: round(1e3 * v['avg_ms'], 1) for n, v in d.get('kernel_cells', {}).items()}
print(json.dumps({'arm': '$tag', 'config': '$CFG', 'frames_per_s': round(d['value'], 2), 'ms_per_video': round(d['ms_per_step'], 2), 'dit_step_ms': round(d.get('dit_step_ms', 0), 3), 'linear_ms': round(d['roofline_class']['ms_total'], 2), 'cells_avg_us': c}))" >> $J; }
CFG=c2
run new A=1
run old LTXHIP_LIB=$R/tools/variants/libltxhip_oldgelu.so
run new-again A=1
run old-again LTXHIP_LIB=$R/tools/variants/libltxhip_oldgelu.so
CFG=c1
run new A=1
run old LTXHIP_LIB=$R/tools/variants/libltxhip_oldgelu.so
cut -c1-420 $J
