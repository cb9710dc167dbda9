#!/bin/bash
# Round 5, third box: the residual prefetch of gemm_asm16's 160-row tiles (bit identity + bench), per-shape GEMM durations from a kernel trace.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_gemm_asm.py tests/test_gpu_tight.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
for i in 1 2; do python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2_$i.json 2> $O/bench.err; python3 -c "
import json; d = json.load(open('$O/bench_c2_$i.json')); print(d['value'], d['dit_step_ms'], d['vae_decode_ms'], {k[:20]: round(v['avg_ms'] * 1e3, 1) for k, v in d['kernels'].items()})"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-prof > $O/stats.log 2>&1
cd $R
python3 tools/summarize_prof.py $O/stats $O/summary 2>&1 | tail -3
head -40 $O/summary.md
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
