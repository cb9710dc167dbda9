#!/bin/bash
# Round 5: VAE norm1 inside the preceding conv2's epilogue + cached timestep modulation: tests, bench, c4 profile (steady state).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5g; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_models.py tests/test_gpu_c2.py tests/test_gpu_c4.py tests/test_gpu_c1.py tests/test_gpu_tight.py tests/test_gpu_determinism.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
J=$O/ab.jsonl; : > $J
run() { tag=$1; shift; env "$@" python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = {n.split(' (')[0].split('/')[0]: {'ms': round(v['ms_total'], 2), 'avg_us': round(1e3 * v['avg_ms'], 1)} for n, v in d.get('kernels', {}).items()}
print(json.dumps({'arm': '$tag', 'config': '$CFG', 'frames_per_s': round(d['value'], 2), 'ms_per_video': round(d['ms_per_step'], 2), 'dit_step_ms': round(d.get('dit_step_ms', 0), 3), 'vae_decode_ms': round(d.get('vae_decode_ms', 0), 2), 'kernels': k}))" >> $J; }
CFG=c2
run default A=1
run "vae_fuse_norm=0" LTX_OPTIONS=vae_fuse_norm=0
run default-again A=1
run "vae_fuse_norm=0 again" LTX_OPTIONS=vae_fuse_norm=0
CFG=c1; run default A=1
CFG=c4; run default A=1
cut -c1-300 $J
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --no-prof"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_stats -- $B > $O/c4_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c4_fetch -- $B > $O/c4_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c4_write -- $B > $O/c4_write.log 2>&1
cd $R && python3 tools/summarize_prof.py $O/c4_stats $O/c4_summary --pmc FETCH_SIZE=$O/c4_fetch --pmc WRITE_SIZE=$O/c4_write 2>&1 | tail -2
sed -n 1,4p $O/c4_summary.md
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
