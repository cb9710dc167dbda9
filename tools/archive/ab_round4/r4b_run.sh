cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
python -m pytest tests/test_gpu_q2fold.py tests/test_gpu_pins.py tests/test_gpu_attn_q128.py -x -q > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
python -m pytest tests/test_gpu_c5.py -x -q -k "width" > $O/pytest_c5w.log 2>&1; tail -5 $O/pytest_c5w.log
for V in fold_on fold_off evstream; do
  unset LTX_Q2_FOLD LTX_PROF_KERNEL_EVENTS
  [ $V = fold_off ] && export LTX_Q2_FOLD=0
  [ $V = evstream ] && export LTX_PROF_KERNEL_EVENTS=0
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_$V.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$V.json").read().strip().splitlines()[-1])
print("$V", round(d["value"],1), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"],4), "class", round(d["roofline_class"]["frac"],4), "attn", round(d["roofline_self_attention"]["avg_launch_ms"]*1e3,1), {k[:28]: round(v["ms_total"],2) for k,v in d["kernels"].items()})
PY
done
unset LTX_Q2_FOLD LTX_PROF_KERNEL_EVENTS
python tools/vs_library.py > $O/gemm_vs_library.jsonl 2> $O/vs.err; cat $O/gemm_vs_library.jsonl
python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; tail -3 $O/pytest_all.log
