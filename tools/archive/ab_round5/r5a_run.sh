#!/bin/bash
# Round 5, first box: tests of the new paths, same-box A/B of (a) cross attention over the valid keys and (b) the frames-fastest
# conv tile order, the conv class's L2-miss traffic under both orders, and the one full C2 oracle run on the box's host cores.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5a; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_xattn_compact.py tests/test_gpu_q2fold.py tests/test_gpu_ops.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
J=$O/ab.jsonl; : > $J
run() { tag=$1; shift; env "$@" python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = {n.split(' (')[0].split('/')[0]: {'ms': round(v['ms_total'], 2), 'avg_us': round(1e3 * v['avg_ms'], 1), 'rate': round(v.get('TFLOP/s', v.get('GB/s', 0)))} for n, v in d.get('kernels', {}).items()}
print(json.dumps({'arm': '$tag', 'config': '$CFG', 'frames_per_s': round(d['value'], 2), 'ms_per_video': round(d['ms_per_step'], 2), 'dit_step_ms': round(d.get('dit_step_ms', 0), 3), 'vae_decode_ms': round(d.get('vae_decode_ms', 0), 2), 'kernels': k}))" >> $J; }
CFG=c2
run default A=1
run "LTX_XATTN_COMPACT=0 (all 128 text keys multiplied)" LTX_XATTN_COMPACT=0
run "LTX_CONV_HALO_ORDER=0 (patch positions fastest)" LTX_CONV_HALO_ORDER=0
run default-again A=1
run "LTX_XATTN_COMPACT=0 again" LTX_XATTN_COMPACT=0
run "LTX_CONV_HALO_ORDER=0 again" LTX_CONV_HALO_ORDER=0
CFG=c1
run default A=1
run "LTX_XATTN_COMPACT=0" LTX_XATTN_COMPACT=0
cut -c1-420 $J
# conv traffic: FETCH / WRITE passes under both orders (program directly after --)
B="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-prof"
cd /tmp && export TMPDIR=/tmp
for ord in 1 0; do
  export LTX_CONV_HALO_ORDER=$ord
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_ord$ord -- $B > $O/fetch_ord$ord.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_ord$ord -- $B > $O/write_ord$ord.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ord$ord -- $B > $O/stats_ord$ord.log 2>&1
done
unset LTX_CONV_HALO_ORDER
cd $R
for ord in 1 0; do python3 tools/summarize_prof.py $O/stats_ord$ord $O/summary_ord$ord --pmc FETCH_SIZE=$O/fetch_ord$ord --pmc WRITE_SIZE=$O/write_ord$ord 2>&1 | tail -3; done
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
timeout 1500 python3 tools/oracle_full_c2.py $O/oracle_c2_full.json 16 > $O/oracle.log 2>&1; tail -2 $O/oracle.log
du -sh $O
