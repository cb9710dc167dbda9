cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
python -m pytest tests/test_gpu_determinism.py tests/test_gpu_c1.py tests/test_gpu_ops.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
export SMALLM_MS=384,128
for cfg in "plain:256,16,4" "sc1:256,16,4" "sc1:256,8,4" "sc1:512,8,4" "sc1:512,8,8" "sc1:512,4,8" "sc1:768,4,8"; do
  proto=${cfg%%:*}; rule=${cfg##*:}
  if [ $proto = plain ]; then export LTX_GEMM_SPLIT_PLAIN=1; else unset LTX_GEMM_SPLIT_PLAIN; fi
  export LTX_GEMM_SPLIT_SMALL=$rule
  echo "== $cfg" >> $O/sweep.txt
  python tools/small_m_probe.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['M'], d['case'], d['us'], d['plan'])" >> $O/sweep.txt
done
cat $O/sweep.txt
