#!/bin/bash
# Round 5: conv_halo's persistent tile loop (next tile's first pieces under the last group) - conv tests, then C2 / C4 against the previous build.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5r; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_tight.py tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_determinism.py tests/test_gpu_gemm_asm.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E " passed| failed|^FAILED|^E " $O/pytest.log | tail -6
J=$O/ab.jsonl; : > $J
run() { tag=$1; shift; env "$@" python3 $R/bench.py --config $CFG --steps 5 --warmup 1 --no-cpu-baseline --no-batched 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = {n: [round(v['ms_total'], 2), round(1e3 * v['avg_ms'], 1)] for n, v in d.get('kernel_cells', {}).items() if 'conv' in n}
print(json.dumps({'arm': '$tag', 'config': '$CFG', 'frames_per_s': round(d['value'], 2), 'ms_per_video': round(d['ms_per_step'], 2), 'vae_decode_ms': round(d.get('vae_decode_ms', 0), 2), 'conv_cells_ms_avgus': c}))" >> $J; }
CFG=c2
run new A=1
run old LTXHIP_LIB=$R/tools/variants/libltxhip_prev.so
run "new, gemm_off=halo_pers" LTX_OPTIONS=gemm_off=halo_pers
run new-again A=1
run old-again LTXHIP_LIB=$R/tools/variants/libltxhip_prev.so
CFG=c4
run new A=1
run old LTXHIP_LIB=$R/tools/variants/libltxhip_prev.so
cut -c1-400 $J
