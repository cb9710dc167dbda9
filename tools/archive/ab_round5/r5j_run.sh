#!/bin/bash
# Round 5: full GPU suite after the explicit-fma epilogues + deferred ff2; C1 / C2 bench A/B.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5j; mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -8 $O/pytest.log
J=$O/ab.jsonl; : > $J
run() { tag=$1; shift; env "$@" python3 $R/bench.py --config $CFG --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = {n.split(' (')[0].split('/')[0]: {'ms': round(v['ms_total'], 2), 'avg_us': round(1e3 * v['avg_ms'], 1)} for n, v in d.get('kernels', {}).items()}
print(json.dumps({'arm': '$tag', 'config': '$CFG', 'frames_per_s': round(d['value'], 2), 'ms_per_video': round(d['ms_per_step'], 2), 'dit_step_ms': round(d.get('dit_step_ms', 0), 3), 'vae_decode_ms': round(d.get('vae_decode_ms', 0), 2), 'plans': d.get('gemm_plans'), 'kernels': k}))" >> $J; }
CFG=c1
run default A=1
run "ff2_defer=0" LTX_OPTIONS=ff2_defer=0
run "x: gemm_split_ring=0 is not available in the shipped build; default-again" A=1
CFG=c2; run default A=1
cut -c1-400 $J
python3 tools/t5_time.py 2>/dev/null | tail -1
