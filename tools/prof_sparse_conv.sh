cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sc -- python $R/bench.py --only-headline --no-configs --steps 30 --warmup 5 --inflight 1 > /dev/null 2>&1
f=$(find /tmp/prof_sc -name "*kernel_stats.csv" | head -1); echo $f
python - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]: print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1))
for r in rows:
    if "sparse" in r["Name"] or "scatter" in r["Name"] or "x3p<64>" in r["Name"]: print("**", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
