cd /tmp && export TMPDIR=/tmp
for la in 1 4; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_la$la -o la$la -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --lookahead $la --no-cpu-baseline > /tmp/prof_la$la.log 2>&1
f=$(find /tmp/prof_la$la -name "*kernel_stats.csv" | head -1)
python - <<PY > $GRAFT_REPO_ROOT/gpurun_out/la${la}_kernel_stats.txt
import csv
rows = list(csv.DictReader(open("$f")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:28]:
    print("%-90s calls %7s total %9.2f ms avg %9.1f us %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
tail -1 /tmp/prof_la$la.log > $GRAFT_REPO_ROOT/gpurun_out/prof_la${la}_bench.json
done
