R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c1prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/bench.py --config c1 --steps 3 --warmup 1 --no-cpu-baseline --no-prof > $O/log 2>&1
f=$(find $O/st -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print(r["Name"][:120], r["Calls"], round(float(r["TotalDurationNs"])/1e6,2), round(float(r["AverageNs"])/1e3,2), round(100*float(r["TotalDurationNs"])/tot,1))
PY
grep -o '"ms_per_step": [0-9.]*' $O/log
rm -rf $O/st
