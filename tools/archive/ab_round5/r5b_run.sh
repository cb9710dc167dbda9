#!/bin/bash
# Round 5, second box: cross attention with two units per wave iteration; the K = N = 2048 linear layers under in-kernel stamps.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
timeout 600 python3 -m pytest tests/test_gpu_xattn_compact.py tests/test_gpu_q2fold.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench.err; python3 -c "
import json; d = json.load(open('$O/bench_c2.json')); print(d['value'], d['dit_step_ms'], d['vae_decode_ms'], {k[:20]: round(v['avg_ms'] * 1e3, 1) for k, v in d['kernels'].items()})"
python3 tools/gemm2048_trace.py run > $O/gemm2048_trace.jsonl 2> $O/trace.err; cat $O/gemm2048_trace.jsonl; tail -3 $O/trace.err
