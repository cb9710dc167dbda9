cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python -m pytest tests -m gpu -x -q > gpurun_out/r4a/pytest.log 2>&1; tail -3 gpurun_out/r4a/pytest.log
python tools/attn_r2_ab.py run attn_r2 attn_3f08 > gpurun_out/r4a/attn_ab.jsonl 2> gpurun_out/r4a/attn_ab.err; cat gpurun_out/r4a/attn_ab.jsonl | cut -c1-400
for i in 1 2; do
for L in head attn_r2; do
  if [ $L = head ]; then unset LTXHIP_LIB; else export LTXHIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libltxhip_$L.so; fi
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r4a/bench_${L}_$i.json 2>> gpurun_out/r4a/bench.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4a/bench_${L}_$i.json").read().strip().splitlines()[-1])
print("$L", d["value"], d.get("roofline_self_attention"), d.get("stages_ms"))
PY
done; done
