#!/bin/bash
# Round 5: kernel-trace + FETCH_SIZE / WRITE_SIZE passes of the configs that had no committed profile (c3, c4, c5), the idle-gap
# report of c3's trace, and the two-videos-in-flight measurement.  Outputs: gpurun_out/r5f/<cfg>_summary.{md,json}, gap_c3.txt.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in c3 c4 c5; do
  B="python3 $R/bench.py --config $c --steps 1 --warmup 0 --no-cpu-baseline --no-prof"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_stats -- $B > $O/${c}_stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${c}_fetch -- $B > $O/${c}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${c}_write -- $B > $O/${c}_write.log 2>&1
  (cd $R && python3 tools/summarize_prof.py $O/${c}_stats $O/${c}_summary --pmc FETCH_SIZE=$O/${c}_fetch --pmc WRITE_SIZE=$O/${c}_write 2>&1 | tail -2)
  if [ $c = c3 ]; then (cd $R && python3 tools/gap_report.py $O/c3_stats 10 > $O/gap_c3.txt 2>&1; head -20 $O/gap_c3.txt); fi
  find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
done
cd $R
python3 tools/two_in_flight.py 4 2>/dev/null | tail -1 > $O/two_in_flight.json; cat $O/two_in_flight.json
du -sh $O
