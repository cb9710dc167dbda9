# kernel-trace averages of one bench run (per kernel instantiation): gpurun -- 'bash tools/ab_kernels.sh TAG [LTX_OPTIONS]'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-batched"
cd /tmp && export TMPDIR=/tmp
LTX_OPTIONS=$2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- $B > $O/t.log 2>&1
cd $R
f=$(find $O/t -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in [x for x in rows if any(k in x["Name"] for k in ("conv_halo","asm16","attn_q64","qknorm","rownorm"))][:22]:
    print(r["Name"][:100], r["Calls"], round(float(r["TotalDurationNs"])/1e6,2), round(float(r["AverageNs"])/1e3,2))
PY
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
