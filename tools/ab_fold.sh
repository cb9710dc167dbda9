R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6b_fold; mkdir -p $O
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-batched"
cd /tmp && export TMPDIR=/tmp
LTX_OPTIONS=norm_fold=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/f0 -- $B > $O/f0.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/f1 -- $B > $O/f1.log 2>&1
LTX_OPTIONS=norm_fold=2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/f2 -- $B > $O/f2.log 2>&1
cd $R
for a in f0 f1 f2; do echo == $a; f=$(find $O/$a -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in [x for x in rows if "asm16" in x["Name"] or "rownorm" in x["Name"] or "attn_q64" in x["Name"]][:14]:
    print(r["Name"][:110], r["Calls"], round(float(r["TotalDurationNs"])/1e6,2), round(float(r["AverageNs"])/1e3,2))
PY
done
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
