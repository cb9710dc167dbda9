cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c1prof -- python3 $GRAFT_REPO_ROOT/bench.py --config c1 --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $GRAFT_REPO_ROOT/gpurun_out/c1prof.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/summarize_prof.py gpurun_out/c1prof gpurun_out/c1prof_summary | head -50
find gpurun_out/c1prof -name "*.csv" -size +1M -delete
