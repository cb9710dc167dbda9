cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
python -m pytest tests/test_gpu_q2fold.py -x -q > $O/pytest_q2.log 2>&1; tail -6 $O/pytest_q2.log
python -m pytest tests/test_gpu_c5.py -x -q -k "width" > $O/pytest_c5w.log 2>&1; tail -4 $O/pytest_c5w.log
python - <<'PY' > $O/q2_micro.json
import sys, os, json, math, torch
sys.path.insert(0, "candle-video_amd"); sys.path.insert(0, "tools")
import ltxhip
from microbench import timeit
M, N, K = 4992, 2048, 2048
x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
res = {"plain": [], "rowsq": [], "standalone_after": []}
for r in range(5):
    res["plain"].append(timeit(lambda: ltxhip.ops.linear(x, w, b), iters=50, warm=5) * 1e3)
    res["rowsq"].append(timeit(lambda: ltxhip.ops.linear_rowsq(x, w, b), iters=50, warm=5) * 1e3)
y = ltxhip.ops.linear(x, w, b)
for r in range(5): res["standalone_after"].append(timeit(lambda: ltxhip.ops.rowsq(y), iters=50, warm=5) * 1e3)
print(json.dumps({k: round(sorted(v)[len(v)//2], 2) for k, v in res.items()}))
PY
cat $O/q2_micro.json
for V in fold_on fold_off fold_on2 fold_off2; do
  unset LTX_Q2_FOLD
  case $V in fold_off*) export LTX_Q2_FOLD=0;; esac
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_$V.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$V.json").read().strip().splitlines()[-1])
print("$V", round(d["value"],1), {k[:28]: round(v["ms_total"],2) for k,v in d["kernels"].items()}, round(d["kernel_cells"]["gemm_asm16_kernel [linear]"]["ms_total"],2))
PY
done
