# same-box A/B of two builds by kernel-trace averages: gpurun -- 'bash tools/ab_lib.sh TAG tools/variants/libltxhip_X.so KERNEL_SUBSTR...'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; V=$R/$2; shift 2; mkdir -p $O
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-batched"
cd /tmp && export TMPDIR=/tmp
for rnd in 1 2; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/head$rnd -- $B > $O/head$rnd.log 2>&1
LTXHIP_LIB=$V rocprofv3 --kernel-trace --stats --output-format csv -d $O/var$rnd -- $B > $O/var$rnd.log 2>&1
done
cd $R
for a in head1 var1 head2 var2; do f=$(find $O/$a -name "*kernel_stats.csv" | head -1); python3 - "$f" "$a" "$@" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r["Name"] for k in sys.argv[3:]): print(sys.argv[2], r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3,2))
PY
done
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
