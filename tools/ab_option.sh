# A/B of one option on the C2 bench under rocprofv3 --kernel-trace: tools/ab_option.sh NAME VAL_A VAL_B [kernel-name filter]
# (run on the GPU box: gpurun -- 'bash tools/ab_option.sh rope_gen 0 1 qknorm'); prints the matching kernels' calls / total ms / avg us per arm
R=$GRAFT_REPO_ROOT; N=$1; O=$R/gpurun_out/ab_$N; mkdir -p $O; F=${4:-.}
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-batched"
cd /tmp && export TMPDIR=/tmp
for v in $2 $3 $2 $3; do
  LTX_OPTIONS=$N=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O/v$v -- $B > $O/v$v.log 2>&1
  echo "== $N=$v  $(grep -o '"ms_per_step": [0-9.]*' $O/v$v.log)"
  f=$(find $O/v$v -name "*kernel_stats.csv" | head -1); python3 - "$f" "$F" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in [x for x in rows if re.search(sys.argv[2], x["Name"])][:8]:
    print("  ", r["Name"][:100], r["Calls"], round(float(r["TotalDurationNs"])/1e6,2), round(float(r["AverageNs"])/1e3,2))
PY
  rm -rf $O/v$v
done
