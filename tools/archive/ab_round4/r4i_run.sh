cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; mkdir -p $O
python -m pytest tests/test_gpu_q2fold.py tests/test_gpu_ops.py -x -q > $O/pytest.log 2>&1; tail -12 $O/pytest.log
for i in 1 2; do for V in presum_on presum_off; do
  unset LTX_NORM_PRESUM; [ $V = presum_off ] && export LTX_NORM_PRESUM=0
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_${V}_$i.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench_${V}_$i.json").read().strip().splitlines()[-1])
print("$V", round(d["value"],1), "dit_step_ms", round(d["dit_step_ms"],3), {k[:22]: round(v["ms_total"],2) for k,v in d["kernels"].items()})
PY
done; done
unset LTX_NORM_PRESUM
python -m pytest tests/test_gpu_c2.py tests/test_gpu_c1.py tests/test_gpu_determinism.py -x -q > $O/pytest2.log 2>&1; tail -5 $O/pytest2.log
