#!/bin/bash
# Round 5, final state: full GPU suite, the round's profile passes (tools/profile_round.sh), the C1 kernel trace, and bench.py's N = 2 path
# walked on one GPU (LTX_BENCH_SAME_GPU=1, gloo: the code path, not a scaling number).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5final; mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; grep -E "passed|failed" $O/pytest.log | tail -2
bash tools/profile_round.sh r5final > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
bash tools/c1prof.sh > /dev/null 2>&1; cp gpurun_out/c1prof_summary.md $O/c1_summary.md; cp gpurun_out/c1prof_summary.json $O/c1_summary.json
LTX_BENCH_SAME_GPU=1 LTX_BENCH_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_n2_same_gpu.json 2> $O/bench_n2.err; echo "n2 rc=$?"; tail -c 600 $O/bench_n2_same_gpu.json | cut -c1-600
