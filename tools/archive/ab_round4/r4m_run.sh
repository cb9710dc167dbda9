cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash tools/profile_round.sh r4m > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
