python tools/profile_host.py 3 4 > gpurun_out/host_profile_x4.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x4 -o x4 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --sample-step 4 --no-cpu-baseline > /tmp/prof_x4.log 2>&1
f=$(find /tmp/prof_x4 -name "*kernel_stats.csv" | head -1)
python - <<PY > $GRAFT_REPO_ROOT/gpurun_out/x4_kernel_stats.txt
import csv
rows = list(csv.DictReader(open("$f")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:25]:
    print("%-90s calls %7s total %9.2f ms avg %9.1f us %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
