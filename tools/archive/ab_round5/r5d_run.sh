#!/bin/bash
# Round 5, fourth box: the whole GPU suite after the option refactor (no getenv in launch paths; experiments compiled out), the
# experiment-only tests against the experiment build, and a bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5d; mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -15 $O/pytest.log
LTXHIP_LIB=$R/tools/variants/libltxhip_exp.so timeout 900 python3 -m pytest tests/test_gpu_attn_q64.py tests/test_gpu_gemm_asm.py -q -m gpu > $O/pytest_exp.log 2>&1; echo "pytest(exp) rc=$?" | tee -a $O/pytest_exp.log; tail -4 $O/pytest_exp.log
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench.err; python3 -c "
import json; d = json.load(open('$O/bench_c2.json')); print(d['value'], d['dit_step_ms'], d['vae_decode_ms'], {k[:20]: round(v['avg_ms'] * 1e3, 1) for k, v in d['kernels'].items()}, d['gemm_plans'])"
