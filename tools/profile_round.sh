#!/bin/bash
# One round's measurement evidence, run on the GPU box:  gpurun -- 'bash tools/profile_round.sh r2b'
#   rocprofv3 kernel trace + three PMC passes (separate runs, as MI355X_MICROARCH.md prescribes) of the headline bench
#   command, condensed by tools/summarize_prof.py; the un-profiled bench line; one line per bench config (c1..c5).
# Outputs land in gpurun_out/<tag>/ (scratch); copy summary.{md,json}, bench_n1.json and bench_modes.jsonl into profiles/.
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-batched"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- $B > $O/sq.log 2>&1
cd $R
python3 tools/summarize_prof.py $O/stats $O/summary --pmc FETCH_SIZE=$O/fetch --pmc WRITE_SIZE=$O/write --sq $O/sq | tail -30
python3 bench.py > $O/bench_n1.json 2> $O/bench.err; tail -c 400 $O/bench_n1.json
: > $O/bench_modes.jsonl
for c in c1 c2 c3 c4 c5; do python3 bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline >> $O/bench_modes.jsonl 2>> $O/bench_modes.err; done
cut -c1-260 $O/bench_modes.jsonl
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete; du -sh $O
