cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_pins.py tests/test_gpu_q2fold.py tests/test_gpu_attn_q64.py -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python tools/qknorm_ab.py rownorm_prev > $O/qknorm_ab.json 2> $O/qknorm_ab.err; cat $O/qknorm_ab.json
for i in 1 2; do for L in head rownorm_prev; do
  if [ $L = head ]; then unset LTXHIP_LIB; else export LTXHIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libltxhip_$L.so; fi
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_${L}_$i.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench_${L}_$i.json").read().strip().splitlines()[-1])
print("$L", round(d["value"],1), "dit_step_ms", round(d["dit_step_ms"],3), "attn_us", round(d["roofline_self_attention"]["avg_launch_ms"]*1e3,1), "xattn", round(d["kernels"]["attn_cross64_kernel (cross attention)"]["ms_total"],2))
PY
done; done
